#!/usr/bin/env python3
"""Reduce the rocprofv3 output of tools/collect_profiles.sh to small files fit for profiles/:
kernel_stats CSVs are copied as they are, counter CSVs become one JSON of per-kernel means."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out = sys.argv[1]
summ = os.path.join(out, 'summary')
os.makedirs(summ, exist_ok=True)
for name in ('bench_net', 'bench_net_fp32', 'train_step', 'bench_net_stepwise', 'bench_net_cache', 'bench_uniform', 'net128'):
    for f in glob.glob(os.path.join(out, name, '*', '*kernel_stats.csv')):
        shutil.copy(f, os.path.join(summ, f'{name}_kernel_stats.csv'))
    log = os.path.join(out, name + '.log')
    if os.path.exists(log):
        lines = [l for l in open(log, errors='replace').read().splitlines()
                 if l.startswith('{') or l.startswith('10x128') or l.startswith('eager:') or l.startswith('graph:')]
        open(os.path.join(summ, f'{name}_run.txt'), 'w').write('\n'.join(lines) + '\n')
pmc = {}
for d in sorted(glob.glob(os.path.join(out, 'pmc*'))):
    if not os.path.isdir(d):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    pmc[os.path.basename(d)] = {
        k.split('(')[0][:80]: {c: {'launches': len(x), 'mean': sum(x) / len(x)} for c, x in v.items()}
        for k, v in agg.items() if any(s in k for s in ('k_sim_step', 'k_c4net', 'k_c4_move_fused', 'k_tower128', 'k_run_uniform', 'k_stem128', 'k_heads128'))}
json.dump(pmc, open(os.path.join(summ, 'pmc_means.json'), 'w'), indent=1, sort_keys=True)
print(json.dumps(pmc, indent=1, sort_keys=True)[:6000])
