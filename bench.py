#!/usr/bin/env python3
"""Self-play throughput benchmark (BASELINE.json metric: Connect4 self-play positions/s
and MCTS sims/s at 800 sims/move, 8192 concurrent games per GPU).

One "step" = one lock-step move of the hot path: every one of the G game slots on this
rank runs S PUCT simulations (select -> leaf eval -> expand/backup) and plays a move,
so a step produces G positions and G*S simulations per rank.  Inputs are synthetic:
all games start from the empty board, finished games restart at once, network weights
are random-initialised (throughput does not depend on their values).

  python bench.py --gpus N --steps K --warmup W [--model uniform|c4_6x32|c4_10x128]
                  [--games G] [--sims S] [--no-cpu-baseline]

N > 1: one rank per GPU over RCCL.  Either torch.distributed.run launches the ranks (RANK /
WORLD_SIZE / MASTER_* in the environment), or -- `python bench.py --gpus N` on its own -- this
script starts them itself as child processes before it touches a GPU and relays rank 0's line
(replaces the reference's mp.spawn of self-play workers, alphazero/train.py:131-134).  Games
shard across ranks with no data-path collective (scaling: weak); per step the packed weights are
broadcast from rank 0 (when a network is in use) and the finished games' records are gathered
on rank 0, both device to device.  Rank 0 prints one JSON line.

At N = 1 the line also carries, measured after the timed region: `fp32_net` (the same loop with
the fp32-grade network kernels), `agreement` (bf16 production path against the fp32-grade kernel in
the loop on identical random streams), `config4_leaf_eval` (the 10x128 network on 16 384-leaf
batches) and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak
F32_PEAK_TFLOPS = 157.3        # f32 MFMA / vector peak


def bytes_per_sim(d, f):
    """Algorithmic HBM bytes per simulation: root state 24 B + per select level one 112-B
    node read and a 16-B backup read-modify-write + per expanded leaf a 120-B node write,
    32 B of (p, v) and a 24-B leaf state."""
    return 24.0 + 128.0 * d + 176.0 * f


def trunk_flops_per_leaf(blocks, ch, cells=42, planes=2):
    conv = 2 * 9 * ch * ch * cells
    return 2 * blocks * conv + 2 * 9 * planes * ch * cells


def cpu_baseline(model_name, sims, seconds, workers):
    """The CPU oracle (oracle/az_oracle.c, the restatement of the reference's algorithm)
    playing self-play moves, one independent single-game worker per core as `train.py -n N`
    does (alphazero/train.py:131-134), one network call per leaf with one torch thread per
    worker (alphazero/mcts.py:94-107).  Bounded sample: each worker plays whole moves until
    `seconds` have passed."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    with ctx.Pool(workers) as pool:
        res = pool.starmap(_cpu_worker, [(model_name, sims, seconds, w) for w in range(workers)])
    pos = sum(r[0] for r in res)
    sim = sum(r[1] for r in res)
    wall = max(r[2] for r in res)
    return pos / wall, sim / wall, pos, wall


def _cpu_worker(model_name, sims, seconds, wid):
    os.environ['OMP_NUM_THREADS'] = '1'
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import ctypes as C
    import numpy as np
    import oracle as orc
    from alphazero_amd.engine import make_puct_lut
    L = orc.lib()
    lut = orc.make_lut_array(make_puct_lut(sims * 42 + 64))
    cfg = orc.make_config(orc.C4, sims, 4, True, lut)
    if model_name == 'uniform':
        ev = orc.eval_uniform()
    else:
        import torch
        torch.set_num_threads(1)
        from alphazero_amd.models import AlphaZeroC4
        blocks, ch = (6, 32) if model_name == 'c4_6x32' else (10, 128)
        torch.manual_seed(0)
        net = AlphaZeroC4(blocks, ch).eval()

        def fn(game, hist, valid):
            x = torch.from_numpy(orc.render(game, hist)[None])
            m = torch.zeros(1, 7, dtype=torch.bool)
            m[0, valid] = True
            with torch.no_grad():
                lg, v = net(x, m)
                p = torch.softmax(lg[m], 0)
            return p.numpy(), float(v[0, 0])
        ev = orc.eval_from_python(fn)
    rng = orc.Rng()
    L.orc_rng_seed(C.byref(rng), 1000 + wid)
    dir_cb = C.cast(L.orc_rng_dirichlet, orc.DIRICHLET_FN)
    t0 = time.time()
    pos = sim = 0
    while time.time() - t0 < seconds:
        # selfplay.py:12-42, one move at a time so the sample can stop on a move boundary
        hist, tree = [], orc.Node()
        top = tree
        while not orc.terminal(orc.C4, hist) and time.time() - t0 < seconds:
            st = orc.Stats()
            h, n = orc._hist(hist)
            L.orc_run_mcts(C.byref(cfg), h, n, tree.ptr, ev, None, dir_cb, C.byref(rng), C.byref(st))
            if len(hist) < 4:
                ai = tree.sample_action(L.orc_rng_uniform(C.byref(rng)))
            else:
                ai = tree.greedy_action()
            hist.append(orc.valid_actions(orc.C4, hist)[ai])
            tree = tree.traverse(ai)
            tree.kill_siblings()
            pos += 1
            sim += st.sims
        del top
    return pos, sim, time.time() - t0


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this same command as children
    of torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and relay what rank 0 prints.
    Runs before anything in this process has touched a GPU; the parent never does."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def physical_cores():
    """(logical CPUs, physical cores) of the host as the kernel reports them."""
    cores = set()
    try:
        phys = core = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                core = line.split(':')[1].strip()
            elif not line.strip():
                if core is not None:
                    cores.add((phys, core))
                phys = core = None
        if core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return os.cpu_count() or 1, len(cores) or None


def cpu_share():
    """CPUs this job may use: the affinity mask, further limited by OMP_NUM_THREADS when the host sets it
    (a one-GPU slice of a shared machine sees every CPU of the machine but owns only its share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        n = min(n, int(os.environ['OMP_NUM_THREADS']))
    except (KeyError, ValueError):
        pass
    return max(n, 1)


def latest_profile(name_suffix):
    """Newest committed profiles/rNN_<suffix> (the rounds' files sort by name)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_{name_suffix}')))
    return files[-1] if files else None


def live_counter_passes(args):
    """FETCH_SIZE and WRITE_SIZE of this command's kernels, measured now: this script run twice as a child process
    under rocprofv3 --pmc (one counter per pass, no tracing), a few steps each.  Returns {counter: {kernel name: mean
    per launch}} or None when the profiler is missing or a pass fails (the committed passes are used then)."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(prof):
        return None
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['TMPDIR'] = '/tmp'
    res = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='az_pmc_', dir='/tmp')
        cmd = [prof, '--pmc', counter, '--output-format', 'csv', '-d', d, '--', sys.executable, os.path.join(ROOT, 'bench.py'),
               '--no-cpu-baseline', '--no-extras', '--steps', '2', '--warmup', '1', '--preroll', '8', '--games', str(args.games),
               '--sims', str(args.sims), '--mix-moves', str(args.mix_moves), '--model', args.model, '--net-impl', args.net_impl]
        if args.stepwise:
            cmd.append('--stepwise')
        try:
            child = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                     start_new_session=True)
            try:
                rc = child.wait(timeout=90)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(child.pid, signal.SIGKILL)        # the profiler and the run under it: this group only
                child.wait()
                return None
            if rc != 0:
                return None
            agg = collections.defaultdict(list)
            for f in glob.glob(os.path.join(d, '*', '*counter_collection.csv')):
                for row in csv.DictReader(open(f)):
                    if row['Counter_Name'] == counter:
                        agg[row['Kernel_Name']].append(float(row['Counter_Value']))
            if not agg:
                return None
            res[counter] = {k: sum(v) / len(v) for k, v in agg.items()}
        except (OSError, subprocess.SubprocessError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return res


def measure_agreement(dev, side_a, side_b, S, note, games=256, moves=12, seed=77):
    """Two evaluators in the same self-play (alphazero/mcts.py:94-107 evaluates in fp32): `games` Connect Four
    games from the empty board on identical Dirichlet / sampling streams (same engine seed: the device RNG is
    a function of seed, game and move).  A side is (network, one_launch): one_launch = all simulations of a
    move in the one-launch kernel, else the stepwise loop.  A (game, move) pair is compared while the two runs
    have played the same moves so far."""
    import numpy as np
    import torch
    from alphazero_amd import engine as E
    engs = [E.SelfPlayEngine(E._lib.GAME_C4, games, S, exploration_depth=4, reuse_search_tree=True,
                             alpha=0.3, epsilon=0.25, seed=seed, device=dev.index) for _ in range(2)]
    same = np.ones(games, bool)
    compared = equal_moves = equal_visits = 0
    l1_sum, l1_max = 0.0, 0.0
    for mv in range(moves):
        out = []
        for eng, (net, one_launch) in zip(engs, (side_a, side_b)):
            bb, info = eng.begin_move()
            p, _ = net(bb, info)
            eng.apply_root(p)
            if one_launch:
                eng.move_fused(net)
            else:
                bb, info = eng.select()
                for s_ in range(S):
                    p, v = net(bb, info)
                    if s_ < S - 1:
                        bb, info = eng.sim_step(p, v)
                    else:
                        eng.expand_backup(p, v)
            a, ne, vis, w = eng.choose_and_advance()
            out.append((a.cpu().numpy().copy(), vis.cpu().numpy().astype(np.float64)))
        (a0, v0), (a1, v1) = out
        live = same & (a0 >= 0) & (a1 >= 0)
        if live.any():
            pi0 = v0[live] / np.maximum(v0[live].sum(1, keepdims=True), 1)
            pi1 = v1[live] / np.maximum(v1[live].sum(1, keepdims=True), 1)
            l1 = np.abs(pi0 - pi1).sum(1)
            compared += int(live.sum())
            equal_moves += int((a0[live] == a1[live]).sum())
            equal_visits += int((v0[live] == v1[live]).all(1).sum())
            l1_sum += float(l1.sum())
            l1_max = max(l1_max, float(l1.max()))
        same &= a0 == a1
    for eng in engs:
        assert eng.error_flags() == 0
        eng.close()
    return {'games': games, 'sims_per_move': S, 'moves_per_game': moves, 'positions_compared': compared,
            'moves_equal_frac': equal_moves / max(compared, 1),
            'visit_vectors_identical_frac': equal_visits / max(compared, 1),
            'mean_l1_pi': l1_sum / max(compared, 1), 'max_l1_pi': l1_max,
            'games_identical_to_the_end': int(same.sum()), 'note': note}


def config4_leaf_eval(dev, iters=10, n=16384):
    """BASELINE config 4: the 10x128 Connect Four network on batches of 16 384 random legal positions
    (uniform random playouts of depth U[0,30]), bf16 MFMA; time per batch from HIP events."""
    import numpy as np
    import torch
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import C4Net128
    from alphazero_amd.games.connectfour import ConnectFour
    rng = np.random.RandomState(0)
    hists = []
    while len(hists) < 2048:                      # 2 048 distinct playouts, repeated to fill the batch
        g, depth = ConnectFour(), rng.randint(0, 31)
        while len(g) < depth and not g.terminal:
            g.take_action(int(rng.choice(g.valid_actions)))
        if not g.terminal:
            hists.append(list(g.history))
    bb, _ = E.positions_from_history(E._lib.GAME_C4, [hists[i % len(hists)] for i in range(n)], device=dev)
    torch.manual_seed(0)
    model = AlphaZeroC4(10, 128).to(dev).eval()
    net = C4Net128(model, dev, max_leaves=n)
    for _ in range(3):
        net(bb)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        net(bb)
        b.record()
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in ev) / iters
    fl = trunk_flops_per_leaf(10, 128) * n
    tf = fl / (ms * 1e-3) / 1e12
    return {'workload': f'AlphaZeroC4 10x128 forward, {n} random legal positions per batch', 'ms_per_batch': ms,
            'leaves_per_s': n / (ms * 1e-3), 'kernel': 'k_stem128 + k_tower128 + k_heads128', 'bound': 'mfma',
            'achieved': tf, 'peak': BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / BF16_PEAK_TFLOPS,
            'flops_per_leaf': trunk_flops_per_leaf(10, 128), 'dtype': 'bf16 MFMA, f32 accumulate'}


def selfplay_10x128(dev, games=8192, sims=800, steps=2, halves=1, seed=1, blocks=10, mix_moves=40, record_sink=None):
    """Self-play with the 10x128 network of BASELINE config 4 (alphazero/models.py:70-87 at that size inside
    alphazero/selfplay.py:24-38): the stepwise loop, one search launch and one network evaluation (leaf list, stem,
    tower, heads) per simulation.  halves = 2 splits the games over two engines on two HIP streams, so that one
    half's search step and small network kernels can run under the other half's tower; the halves own the game
    uids of the one-engine run (same device RNG streams), so the finished games are the same ones."""
    import torch
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import C4Net128
    torch.manual_seed(0)
    model = AlphaZeroC4(blocks, 128).to(dev).eval()
    per = games // halves
    engs = [E.SelfPlayEngine(E._lib.GAME_C4, per, sims, exploration_depth=4, reuse_search_tree=True, alpha=0.3,
                             epsilon=0.25, seed=seed, device=dev.index, first_game_uid=h * per, uid_stride=games)
            for h in range(halves)]
    nets = [C4Net128(model, dev, max_leaves=per) for _ in range(halves)]
    torch.cuda.synchronize()                       # buffers made on the current stream, used on the halves' streams
    streams = [torch.cuda.Stream(device=dev) for _ in range(halves)] if halves > 1 else [torch.cuda.current_stream(dev)]
    tower_ev = []

    def step(measure=False):
        st = [None] * halves
        for h, (eng, net) in enumerate(zip(engs, nets)):
            with torch.cuda.stream(streams[h]):
                bb, info = eng.begin_move()
                p, _ = net(bb, info)
                eng.apply_root(p)
                st[h] = eng.select()
        for s_ in range(sims):
            for h, (eng, net) in enumerate(zip(engs, nets)):
                with torch.cuda.stream(streams[h]):
                    sample = measure and h == 0 and s_ % 32 == 7
                    if sample:
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record()
                    p, v = net(*st[h])
                    if sample:
                        b.record()
                        tower_ev.append((a, b))
                    if s_ < sims - 1:
                        st[h] = eng.sim_step(p, v)
                    else:
                        eng.expand_backup(p, v)
        for h, eng in enumerate(engs):
            with torch.cuda.stream(streams[h]):
                eng.choose_and_advance()
                raw = eng.drain_finished_bytes()
                if record_sink is not None:
                    record_sink.append(raw.copy())
    for i in range(mix_moves):                     # spread the games' ages first (uniform prior: cheap)
        for h, eng in enumerate(engs):
            with torch.cuda.stream(streams[h]):
                eng.begin_move_uniform()
                eng.run_sims_uniform(sims)
                eng.choose_and_advance()
                if i % 2 == 1:
                    raw = eng.drain_finished_bytes()
                    if record_sink is not None:
                        record_sink.append(raw.copy())
    step()
    st0 = [eng.stats() for eng in engs]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / steps
    st1 = [eng.stats() for eng in engs]
    d_sims = sum(b['sims'] - a['sims'] for a, b in zip(st0, st1))
    leaves = sum(b['leaf_evals'] - a['leaf_evals'] for a, b in zip(st0, st1))
    assert d_sims == games * sims * steps and all(b['errors'] == 0 for b in st1)
    net_ms = sum(a.elapsed_time(b) for a, b in tower_ev) / max(len(tower_ev), 1)
    per_launch = leaves / (steps * sims * halves)
    tf = per_launch * trunk_flops_per_leaf(blocks, 128) / (net_ms * 1e-3) / 1e12 if halves == 1 else None
    for eng in engs:
        eng.close()
    return {'value': games / dt, 'unit': 'positions/s', 'sims_per_s': games * sims / dt, 'ms_per_step': 1e3 * dt,
            'steps': steps, 'engines_on_streams': halves, 'leaves_per_network_call': per_launch,
            'network_call_ms': net_ms if halves == 1 else None,
            'network_tflops_in_loop': tf, 'network_frac_of_bf16_peak': tf / BF16_PEAK_TFLOPS if tf else None}


def config2_selfplay(dev, fused, games=1024, sims=200, steps=12):
    """BASELINE config 2: 1 024 concurrent Connect Four games, 200 simulations per move, on one GPU; the same path
    as the headline line (one az_c4_move_fused launch per move, four games per compute unit here)."""
    import torch
    from alphazero_amd import engine as E
    eng = E.SelfPlayEngine(E._lib.GAME_C4, games, sims, exploration_depth=4, reuse_search_tree=True,
                           alpha=0.3, epsilon=0.25, seed=1, device=dev.index)

    def step():
        bb, info = eng.begin_move()
        p, _ = fused(bb, info)
        eng.apply_root(p)
        eng.move_fused(fused)
        eng.choose_and_advance()
        eng.drain_finished_bytes()
    for i in range(120):                          # spread the games' ages first (uniform prior: cheap)
        eng.begin_move_uniform()
        eng.run_sims_uniform(sims)
        eng.choose_and_advance()
        if i % 2 == 1:
            eng.drain_finished()
    for _ in range(10):
        step()
    st0 = eng.stats()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / steps
    st1 = eng.stats()
    assert st1['sims'] - st0['sims'] == games * sims * steps and st1['errors'] == 0
    eng.close()
    return {'workload': f'connect4 self-play, {sims} sims/move, {games} games on one GPU, model c4_6x32',
            'value': games / dt, 'unit': 'positions/s', 'sims_per_s': games * sims / dt, 'ms_per_step': 1e3 * dt,
            'steps': steps, 'launches': 'one per move (az_c4_move_fused)',
            'leaf_eval_fraction': (st1['leaf_evals'] - st0['leaf_evals']) / (games * sims * steps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--model', default='c4_6x32', choices=['uniform', 'c4_6x32', 'c4_10x128'])
    ap.add_argument('--net-impl', default='hip', choices=['torch', 'hip'])
    ap.add_argument('--net-precision', default='bf16', choices=['bf16', 'fp32'],
                    help="c4_6x32 with --net-impl hip: 'fp32' times the one-launch kernel with the fp16-pair network "
                         "(outputs within 1e-5 of the fp32 module) instead of the bf16 one")
    ap.add_argument('--games', type=int, default=8192)
    ap.add_argument('--sims', type=int, default=800)
    ap.add_argument('--preroll', type=int, default=24,
                    help='untimed lock-step moves played before warmup so that the game slots are spread '
                         'over opening / middle game / endgame as in a long-running job')
    ap.add_argument('--mix-moves', type=int, default=160,
                    help='untimed lock-step moves with the in-kernel uniform prior played first: all slots start '
                         'at move 0, and only after several game lengths are their ages spread the way a '
                         'long-running job has them (cheap: ~12 ms per move)')
    ap.add_argument('--node-capacity', type=int, default=0, help='nodes per game pool (0 = never-free bound)')
    ap.add_argument('--allow-pool-errors', action='store_true', help='tuning only')
    ap.add_argument('--eval-cache-log2', type=int, default=0,
                    help='log2 entries of the engine evaluation cache (0 = off, the default: every leaf goes '
                         'through the network as in the reference); the cache is invalidated every step, as '
                         'a weight update per move would')
    ap.add_argument('--stepwise', action='store_true',
                    help='one network launch and one search launch per simulation instead of az_c4_move_fused '
                         '(all simulations of a move in one launch, a workgroup owns its games), which is the '
                         'default for the <= 32-channel HIP network up to 16384 games')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-live-traffic', action='store_true',
                    help='take roofline.traffic from the newest committed profiles/rNN_pmc_means.json instead of measuring it '
                         'now (two child runs of this command under rocprofv3 --pmc)')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the extra measurements of the N = 1 line, made after the timed region (fp32 network, agreement, '
                         'BASELINE configs 2, 4 and 5)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from alphazero_amd import engine as E
    from alphazero_amd.distributed import broadcast_packed, gather_records_device

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # AZ_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend -- a rehearsal of the N > 1 code
    # path on a one-GPU box (RCCL refuses two ranks on one device); never a measurement
    rehearse = os.environ.get('AZ_BENCH_REHEARSE') == '1'
    if rehearse:
        local = 0
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher started {world} ranks')
    # AZ_BENCH_FORCE_DIST=1: run the multi-rank code path (RCCL process group, weight broadcast, record gather,
    # barrier-bracketed timing) even with a single rank -- a check of the collectives on a one-GPU box
    multi = world > 1 or os.environ.get('AZ_BENCH_FORCE_DIST') == '1'
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        torch.cuda.set_device(local)
        if rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local if multi else 0)
    G, S = args.games, args.sims

    eng = E.SelfPlayEngine(E._lib.GAME_C4, G, S, exploration_depth=4, reuse_search_tree=True,
                           alpha=0.3, epsilon=0.25, seed=0, device=dev.index,
                           first_game_uid=rank * G, uid_stride=world * G, node_capacity=args.node_capacity)

    net = None
    if args.model != 'uniform':
        from alphazero_amd.models import AlphaZeroC4
        from alphazero_amd.mcts import evaluate_leaves
        # alphazero/config/connectfour.yaml:4-9 ships 6x32; 10x128 is the larger leaf-eval configuration
        blocks, ch = (6, 32) if args.model == 'c4_6x32' else (10, 128)
        torch.manual_seed(0)
        net = AlphaZeroC4(blocks, ch).to(dev).eval()
        if args.net_impl == 'hip':
            from alphazero_amd.net import C4Net128, FusedC4Net, FusedC4NetHX
            if ch <= 32:
                fused = FusedC4NetHX(net, dev) if args.net_precision == 'fp32' else FusedC4Net.from_module(net, dev)
            else:
                fused = C4Net128(net, dev, max_leaves=G)
            evaluate = lambda bb, info: fused(bb, info)
        else:
            evaluate = lambda bb, info: evaluate_leaves(eng.game, net, bb, info)

    use_cache = args.eval_cache_log2 > 0 and net is not None and args.net_impl == 'hip'
    fused_move = (net is not None and args.net_impl == 'hip' and args.model == 'c4_6x32' and not args.stepwise
                  and G <= 16384)
    stepwise_now = [not fused_move]       # the untimed sampling steps after the timed region flip this
    if use_cache:
        eng.enable_eval_cache(args.eval_cache_log2)
    step_no = [0]
    finished_on_dst = [0]
    packed_wire = net is not None and args.net_impl == 'hip' and args.model == 'c4_6x32'
    net_ev, tree_ev, move_ev = [], [], []       # (start, end) HIP event pairs on the launch stream

    def timed(fn, sink, on):
        if not on:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        sink.append((a, b))
        return r

    def one_step(measure):
        if net is None:
            eng.begin_move_uniform()
            timed(lambda: eng.run_sims_uniform(S), tree_ev, measure)
        else:
            if multi and packed_wire:
                # RCCL weight broadcast replaces model_server.latest() (model_server.py:32-40): the
                # publishing rank packs its module once, the kernel-ready buffer travels, receivers
                # evaluate from where it lands
                if rank == 0:
                    fused.load_module(net)
                broadcast_packed(fused, step_no[0] + 1, src=0)
            elif multi:
                from alphazero_amd.distributed import broadcast_model
                broadcast_model(net, 1, src=0)
                if args.net_impl == 'hip':
                    fused.load_module(net)
            if use_cache:
                step_no[0] += 1
                eng.set_model_version(step_no[0])       # nothing cached under earlier weights is used
            bb, info = eng.begin_move()
            p, _ = evaluate(bb, info)
            eng.apply_root(p)
            if not stepwise_now[0]:
                timed(lambda: eng.move_fused(fused), move_ev, measure)
                eng.choose_and_advance()
                if measure:
                    hand_over()
                return
            bb, info = eng.select()
            for s in range(S):
                sample = measure and (s % 16 == 7)      # every 16th launch carries event pairs
                p, v = timed(lambda: evaluate(bb, info), net_ev, sample)
                if s < S - 1:
                    bb, info = timed(lambda: eng.sim_step(p, v), tree_ev, sample)
                else:
                    eng.expand_backup(p, v)
        eng.choose_and_advance()
        if measure:
            hand_over()

    def hand_over():
        # finished games are handed over every move (selfplay.py:41): on one GPU to the host, as the
        # reference's dataset sink wants them; with several ranks they stay in device memory and are
        # gathered on the replay rank over RCCL (replaces replay_buffer.py:28-47)
        if multi:
            recs, head = eng.take_finished_device()
            got, flags, _ = gather_records_device(recs, head, dst=0)
            assert flags == 0 or args.allow_pool_errors, f'engine error bits {flags}'
            finished_on_dst[0] += int(got.shape[0])
        else:
            eng.drain_finished_bytes()

    def sync_all():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.mix_moves):
        eng.begin_move_uniform()
        eng.run_sims_uniform(S)
        eng.choose_and_advance()
        if i % 2 == 1:
            eng.drain_finished()
    for i in range(args.preroll):
        one_step(False)
        if i % 2 == 1:
            eng.drain_finished()
    for _ in range(args.warmup):
        one_step(False)
    eng.drain_finished()
    st0 = eng.stats()
    hits0 = eng.eval_cache_hits() if use_cache else 0
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(True)
    sync_all()
    elapsed = time.perf_counter() - t0
    st1 = eng.stats()
    hits = (eng.eval_cache_hits() - hits0) if use_cache else 0
    if fused_move and not multi and args.net_precision == 'bf16':
        # untimed, single-GPU runs only: two moves through the per-simulation kernels, for their own durations (the search
        # kernel against HBM, the network kernel against the matrix cores)
        stepwise_now[0] = True
        for _ in range(2):
            one_step(True)
        torch.cuda.synchronize()
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    sims = st1['sims'] - st0['sims']
    evals = st1['leaf_evals'] - st0['leaf_evals']
    d = (st1['select_steps'] - st0['select_steps']) / max(sims, 1)
    f = evals / max(sims, 1)
    assert sims == G * S * args.steps or args.allow_pool_errors, (sims, G * S * args.steps)
    assert st1['errors'] == 0 or args.allow_pool_errors, 'node pool exhausted'

    positions = world * G * args.steps
    out = {
        **({'rehearsal': 'all ranks on one GPU over gloo: not a measurement'} if rehearse else {}),
        'metric': f'Connect4 self-play positions/s ({S} sims/move)',
        'value': positions / elapsed,
        'unit': 'positions/s',
        'sims_per_s': world * sims / elapsed,
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64 PUCT score, f32 W/P, u32 N' + ('' if net is None else (
            '; net trunk fp16-pair operands (three f16 MFMAs per product), f32 accumulate: f32 grade'
            if args.net_precision == 'fp32' and ch <= 32 and args.net_impl == 'hip' else '; net trunk bf16 MFMA, f32 accumulate')),
        'data': 'synthetic',
        'config': {'workload': f'connect4 self-play, {S} sims/move, {G} games/GPU, exploration_depth 4, '
                               f'reuse_search_tree, model {args.model}, {args.mix_moves} age-mixing moves + '
                               f'{args.preroll} pre-roll moves',
                   'games_per_gpu': G, 'sims_per_move': S, 'net': args.model,
                   'launches': 'one per move (az_c4_move_fused)' if fused_move else 'per simulation',
                   'net_impl': None if net is None else args.net_impl,
                   'net_precision': None if net is None else args.net_precision, 'preroll_moves': args.preroll,
                   'mix_moves': args.mix_moves,
                   'mean_select_depth': d, 'leaf_eval_fraction': f,
                   'eval_cache_log2': args.eval_cache_log2 if use_cache else 0,
                   'eval_cache_hit_rate': hits / max(evals, 1),
                   'games_finished': st1['games'] - st0['games'],
                   'exchange': None if not multi else
                   ('per step: broadcast_packed (kernel-ready weights, one in-place broadcast) + '
                    'gather_records_device (records device to device)' if packed_wire else
                    'per step: broadcast_model + gather_records_device'),
                   'records_gathered_on_rank0': finished_on_dst[0] if multi else None},
    }
    if rank == 0:
        def avg_ms(pairs):
            return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)

        # HBM traffic of the kernels of this command: measured now when this is the plain one-GPU line (two child runs
        # under rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in passes of their own), else from the committed passes
        live = None
        if not multi and not args.no_extras and not args.no_live_traffic and net is not None and args.net_impl == 'hip':
            live = live_counter_passes(args)
        out['traffic_source'] = ('two rocprofv3 --pmc passes of this command (FETCH_SIZE, WRITE_SIZE), run by bench.py after the '
                                 'timed region' if live else
                                 'newest committed profiles/rNN_pmc_means.json (counter passes of this command)')

        def pmc_traffic(*kernel_prefixes):
            """HBM bytes per launch from counter passes of this same command (FETCH_SIZE and WRITE_SIZE in KiB,
            separate passes; FETCH_SIZE doubled, the gfx950 correction for 16-byte-per-lane reads): the live
            passes when there are any, else profiles/rNN_pmc_means.json.  A kernel is named by the start of its
            signature, with or without the leading return type."""
            def named(k):
                k = k[5:] if k.startswith('void ') else k
                return any(k.startswith(p[5:] if p.startswith('void ') else p) for p in kernel_prefixes)
            try:
                if live:
                    fe = [v for k, v in live['FETCH_SIZE'].items() if named(k)]
                    wr = [v for k, v in live['WRITE_SIZE'].items() if named(k)]
                    if fe and wr:
                        return (2.0 * fe[0] + wr[0]) * 1024.0
                pm = json.load(open(latest_profile('pmc_means.json')))
                for fk, wk in (('pmc_fetch', 'pmc_write'), ('pmc_stepwise_fetch', 'pmc_stepwise_write')):
                    fe = [v['FETCH_SIZE']['mean'] for k, v in pm.get(fk, {}).items() if named(k)]
                    wr = [v['WRITE_SIZE']['mean'] for k, v in pm.get(wk, {}).items() if named(k)]
                    if fe and wr:
                        return (2.0 * fe[0] + wr[0]) * 1024.0
                return None
            except (OSError, KeyError, IndexError, ValueError, TypeError):
                return None
        bps = bytes_per_sim(d, f)
        if fused_move:
            # the whole move is one launch: its matrix-core work is the leaves evaluated in the move
            ms = avg_ms(move_ev)
            fl = trunk_flops_per_leaf(blocks, ch) * (evals - hits) / args.steps
            tf = fl / (ms * 1e-3) / 1e12
            hx_run = args.net_precision == 'fp32'
            issued = 3.0 if hx_run else 1.0              # MFMAs issued per product
            out['roofline'] = {'kernel': 'k_c4_move_fused<fp16 pairs>' if hx_run else 'k_c4_move_fused', 'bound': 'mfma',
                               'achieved': tf, 'peak': BF16_PEAK_TFLOPS,
                               'unit': 'TFLOP/s', 'frac': issued * tf / BF16_PEAK_TFLOPS, 'mfma_issued_tflops': issued * tf,
                               'traffic': None if hx_run else pmc_traffic('k_c4_move_fused<false>', 'k_c4_move_fused('), 'avg_launch_ms': ms,
                               'flops_per_leaf': trunk_flops_per_leaf(blocks, ch),
                               'leaves_per_launch': (evals - hits) / args.steps,
                               'note': 'all simulations of a move: tree walks and network tasks of different games '
                                       'interleaved on the same waves; the tree walks do no matrix work'}
        if net is not None and args.net_impl == 'hip' and net_ev and tree_ev:
            ms = avg_ms(net_ev)
            # leaves one launch evaluates x flops per leaf; the 128-channel kernels take every leaf of
            # the batch, needed or not, and only the needed ones count as algorithmic work
            f_net = (evals - hits) / max(sims, 1)          # leaves that reach the network
            fl = trunk_flops_per_leaf(blocks, ch) * f_net * G
            tf = fl / (ms * 1e-3) / 1e12
            kname = 'k_c4net_bf16' if ch <= 32 else 'k_stem128 + k_tower128 + k_heads128'
            out['roofline_stepwise_net' if fused_move else 'roofline'] = {
                               'kernel': kname, 'bound': 'mfma', 'achieved': tf, 'peak': BF16_PEAK_TFLOPS,
                               'unit': 'TFLOP/s', 'frac': tf / BF16_PEAK_TFLOPS,
                               'traffic': pmc_traffic('k_c4net_bf16') if ch <= 32 else None,
                               'avg_launch_ms': ms, 'flops_per_leaf': trunk_flops_per_leaf(blocks, ch),
                               'leaves_per_launch': f_net * G}
            ms = avg_ms(tree_ev)
            gbs = bps * G / (ms * 1e-3) / 1e9
            out['roofline_tree'] = {'kernel': 'k_sim_step<GameC4,true,true>', 'bound': 'hbm', 'achieved': gbs,
                                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                                    'traffic': pmc_traffic('void k_sim_step_p<GameC4, 8, true, true>'),
                                    'algorithmic_bytes_per_launch': bps * G,
                                    'avg_launch_ms': ms, 'bytes_per_sim': bps}
        elif net is None and tree_ev:
            ms = avg_ms(tree_ev)
            gbs = bps * G * S / (ms * 1e-3) / 1e9
            out['roofline'] = {'kernel': 'k_run_uniform<GameC4>', 'bound': 'hbm', 'achieved': gbs,
                               'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                               'traffic': None, 'avg_launch_ms': ms, 'bytes_per_sim': bps}
        if not multi and fused_move and not args.no_extras and args.net_precision == 'bf16':
            from alphazero_amd.net import FusedC4NetF32
            f32 = FusedC4NetF32(net, dev)
            def extra(name, fn):
                # a failure in an extra measurement must not cost the line its headline figures
                try:
                    out[name] = fn()
                except Exception as e:                  # noqa: BLE001
                    out[name] = {'error': f'{type(e).__name__}: {e}'}
            from alphazero_amd.net import FusedC4NetHX
            hx = FusedC4NetHX(net, dev)

            def extra_fp32():
                # (a) the same self-play loop at the reference's own network precision (alphazero/mcts.py:94-107
                # evaluates in fp32): the one-launch move kernel with the fp16-pair network (three MFMAs per product,
                # outputs within 1e-5 of the fp32 module), and the stepwise loop with the fp32-MFMA kernel beside it
                def hx_step():
                    bb, info = eng.begin_move()
                    p, _ = hx(bb, info)
                    eng.apply_root(p)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    eng.move_fused(hx)
                    b.record()
                    eng.choose_and_advance()
                    eng.drain_finished_bytes()
                    return a, b

                def f32_step():
                    bb, info = eng.begin_move()
                    p, _ = f32(bb, info)
                    eng.apply_root(p)
                    bb, info = eng.select()
                    for s_ in range(S):
                        p, v = f32(bb, info)
                        if s_ < S - 1:
                            bb, info = eng.sim_step(p, v)
                        else:
                            eng.expand_backup(p, v)
                    eng.choose_and_advance()
                    eng.drain_finished_bytes()
                hx_step()
                st_a = eng.stats()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                evs = [hx_step() for _ in range(4)]
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / 4
                st_b = eng.stats()
                assert st_b['errors'] == 0 and st_b['sims'] - st_a['sims'] == 4 * G * S
                launch_ms = sum(a.elapsed_time(b) for a, b in evs) / 4
                leaves = (st_b['leaf_evals'] - st_a['leaf_evals']) / 4
                tf = leaves * trunk_flops_per_leaf(blocks, ch) / (launch_ms * 1e-3) / 1e12
                f32_step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                f32_step()
                torch.cuda.synchronize()
                dt32 = time.perf_counter() - t1
                return {'value': G / dt, 'unit': 'positions/s', 'sims_per_s': G * S / dt, 'ms_per_step': 1e3 * dt, 'steps': 4,
                        'kernel': 'k_c4_move_fused<fp16 pairs> (az_c4_move_fused_hx: v_mfma_f32_16x16x32_f16, three per '
                                  'product, f32 accumulate; heads in f32)',
                        'roofline': {'bound': 'mfma', 'avg_launch_ms': launch_ms, 'leaves_per_launch': leaves,
                                     'achieved': tf, 'unit': 'TFLOP/s (algorithmic, fp32-grade)',
                                     'mfma_issued_tflops': 3 * tf, 'peak': BF16_PEAK_TFLOPS,
                                     'frac': 3 * tf / BF16_PEAK_TFLOPS,
                                     'note': 'frac = issued fp16 matrix work (3 MFMAs per product) / dense fp16 peak'},
                        'stepwise_f32_mfma': {'value': G / dt32, 'unit': 'positions/s', 'ms_per_step': 1e3 * dt32, 'steps': 1,
                                              'kernel': 'k_c4net_f32 (v_mfma_f32_32x32x2_f32, every leaf) + k_sim_step_p per simulation'},
                        'note': 'network outputs of both kernels are within 1e-5 relative of the reference fp32 net'}

            def extra_cache():
                # (e) the evaluation cache (opt-in; the reference evaluates every leaf, alphazero/mcts.py:139-142): same
                # moves and records bit for bit, 2^24 entries, the model version bumped every move (no reuse across moves)
                eng.enable_eval_cache(24)
                ver = [1000]

                def c_step():
                    ver[0] += 1
                    eng.set_model_version(ver[0])
                    one_step(False)
                    eng.drain_finished_bytes()
                c_step()
                st_a, h_a = eng.stats(), eng.eval_cache_hits()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(4):
                    c_step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / 4
                st_b, h_b = eng.stats(), eng.eval_cache_hits()
                eng.enable_eval_cache(0)
                ev = st_b['leaf_evals'] - st_a['leaf_evals']
                return {'value': G / dt, 'unit': 'positions/s', 'ms_per_step': 1e3 * dt, 'steps': 4, 'log2_entries': 24,
                        'hit_rate': (h_b - h_a) / max(ev, 1),
                        'note': 'az_eval_cache_enable: bit-identical moves and records (tests/test_gpu_cache.py); not part of '
                                'the headline value, which evaluates every leaf like the reference'}
            def extra_config5():
                # (d) BASELINE config 5 on one GPU: the training step on the device-resident replay window, alone
                # and inside the self-play loop (one SGD step and one weight hand-over per move of all games, as the
                # reference's synchronous mode alternates them, alphazero/train.py:86-96)
                from alphazero_amd.config import AlphaZeroConfig
                from alphazero_amd.games import ConnectFour
                from alphazero_amd.replay_buffer import DeviceReplayWindow
                from alphazero_amd.training import TrainingWorker
                tcfg = AlphaZeroConfig(conf={
                    'game': {'module': 'alphazero.games.connectfour', 'name': 'ConnectFour'},
                    'model': {'module': 'alphazero.games.connectfour', 'name': 'AlphaZeroC4',
                              'kwargs': {'num_blocks': blocks, 'channels_per_block': ch}},
                    'loss': {'module': 'alphazero.loss', 'name': 'AlphaZeroLoss'},
                    'training': {'num_steps': 10000, 'batch_size': 128, 'lr': 0.2, 'momentum': 0.9, 'weight_decay': 1e-4,
                                 'lr_schedule_gamma': 0.1, 'lr_schedule': [1000, 2000, 4000]}})   # config/connectfour.yaml:13-20

                class _Publish:
                    def update(self, model, version):
                        self.version = version
                window = DeviceReplayWindow(ConnectFour, 64 * 1024, dev)
                stepwise_now[0] = False

                def play_and_collect():
                    one_step(False)
                    recs, head = eng.take_finished_device()
                    n_rec, flags = head.tolist()
                    assert flags == 0
                    window.add_records_device(recs[:n_rec])
                while len(window) < 8 * 128:
                    play_and_collect()
                torch.manual_seed(0)
                tw = TrainingWorker(tcfg, _Publish(), window, None, dev)
                for _ in range(tw.EAGER_STEPS + 3):           # the ordinary steps, the recording of the step, two replays
                    tw.process_batch()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(100):
                    tw.process_batch()
                torch.cuda.synchronize()
                train_ms = 1e3 * (time.perf_counter() - t1) / 100
                play_and_collect()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                loop_steps = 4
                for _ in range(loop_steps):
                    # the SGD step and the re-pack are queued behind the move before the host waits for the move's
                    # records, so the device goes from one to the other without idling (an idle device drops its clocks
                    # and the 1.5 ms step then takes 5-10 ms: tools/train_loop_probe.py); the step therefore trains on the
                    # window as it was one move earlier
                    one_step(False)
                    v_new = tw.process_batch()
                    fused.load_module(tw.model)                  # the new weights reach the self-play kernels
                    eng.set_model_version(v_new)
                    recs, head = eng.take_finished_device()
                    n_rec, flags = head.tolist()
                    assert flags == 0
                    window.add_records_device(recs[:n_rec])
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / loop_steps
                return {
                    'training_step_ms': train_ms, 'batch_size': 128, 'samples_per_s': 128 / (train_ms * 1e-3),
                    'step': 'az_sample_window + az_c4_train_fwd_bwd (hand-written forward on batch statistics, loss, backward and '
                            f'batch sums in ONE resident launch whose {4 * blocks + 9} steps exchange the batch statistics among the '
                            'workgroups; no vendor convolution, no autograd) + az_sgd_momentum + publish; four launches, replayed '
                            'as one recorded HIP graph',
                    'hand_written_pass': tw._hand is not None,
                    'selfplay_with_training': {'value': G / dt, 'unit': 'positions/s', 'ms_per_step': 1e3 * dt,
                                               'steps': loop_steps,
                                               'per_step': 'one move of every game (records stay on the device) + one SGD step + '
                                                           're-pack of the weights for the self-play kernels'},
                    'window_states': len(window), 'note': 'one GPU; with N ranks the records are gathered on the trainer rank '
                                                          'and the packed weights broadcast (see config.exchange)'}
            stepwise_now[0] = False
            extra('fp32_net', extra_fp32)
            # (b) how often the bf16 production path changes a move or a visit vector against the fp32-grade network
            # (both in the one-launch kernel), and how the two fp32-grade kernels compare with each other
            extra('agreement', lambda: measure_agreement(
                dev, (fused, True), (hx, True), S,
                'bf16 MFMA network vs fp16-pair (fp32-grade) network, both in the one-launch move kernel; same noise and '
                'sampling streams; pairs compared while both runs share the history'))
            extra('agreement_fp32_kernels', lambda: measure_agreement(
                dev, (hx, True), (f32, False), S,
                'fp16-pair network in the one-launch kernel vs az_c4net_f32_forward (fp32 MFMA) in the stepwise loop: two '
                'kernels that both hold 1e-5 against the fp32 module, different summation orders', games=128, moves=8))
            extra('eval_cache', extra_cache)
            # (c) BASELINE configs 2 and 4
            extra('config2_selfplay', lambda: config2_selfplay(dev, fused))
            extra('config4_leaf_eval', lambda: config4_leaf_eval(dev))
            extra('config5_train_loop', extra_config5)

            def extra_10x128():
                # (f) self-play with the 10x128 network (BASELINE config 4's network inside the config-3 loop), one
                # engine and two engines on two streams
                one = selfplay_10x128(dev, G, S, steps=2, halves=1)
                two = selfplay_10x128(dev, G, S, steps=2, halves=2)
                one['workload'] = f'connect4 self-play, {S} sims/move, {G} games/GPU, model c4_10x128, stepwise loop'
                one['two_halves_on_two_streams'] = {k: two[k] for k in ('value', 'ms_per_step', 'leaves_per_network_call')}
                return one
            extra('selfplay_10x128', extra_10x128)
        if not args.no_cpu_baseline and not multi:      # the host-CPU baseline is timed at N = 1 only
            logical, physical = physical_cores()
            workers = cpu_share()
            try:
                pps, sps, npos, wall = cpu_baseline(args.model, S, args.cpu_seconds, workers)
            except Exception as e:                      # noqa: BLE001 (the line keeps its GPU figures)
                out['cpu_baseline'] = {'error': f'{type(e).__name__}: {e}'}
                print(json.dumps(out))
                if multi:
                    dist.destroy_process_group()
                return
            try:        # oracle speed / reference-Python speed, measured in the build container (tools/ref_ratio.py)
                ratio = json.load(open(latest_profile('ref_vs_oracle_cpu.json')))['oracle_over_reference']
            except (OSError, KeyError, ValueError, TypeError):
                ratio = None
            out['cpu_baseline'] = {'value': pps, 'unit': 'positions/s', 'sims_per_s': sps, 'cores': workers,
                                   'host_logical_cpus': logical, 'host_physical_cores': physical,
                                   'cores_note': 'workers = the CPUs this job owns (affinity mask, capped by '
                                                 'OMP_NUM_THREADS where the host sets a per-job share); the '
                                                 'machine-wide counts are given beside it',
                                   'kind': 'port', 'oracle_over_reference_python': ratio,
                                   'reference_python_equivalent': pps / ratio if ratio else None,
                                   'sample': f'{npos} self-play moves ({S} sims each) by the CPU oracle in {wall:.1f} s: '
                                             f'{workers} independent single-game workers, one core and one net call '
                                             f'per leaf each, model {args.model}'}
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
