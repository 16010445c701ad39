"""Shows the stale LDS read that the value head of the fp16-pair network ran into (csrc/az_net_hx_dev.h,
hx_heads_own).  Run it once with the product library and once with the diagnostic build in which the compiler
places the LDS waits itself:

    make -C alphazero_amd/csrc OUT=../variants/hx_partial_wait.so EXTRA=-DHX_PARTIAL_WAIT
    python tools/exp_lds_wait_probe.py
    AZHIP_LIBRARY=alphazero_amd/variants/hx_partial_wait.so python tools/exp_lds_wait_probe.py

It evaluates 256 positions with az_c4net_hx_forward ten times and reports how many values differ from the first run
and how many are further than 1e-6 from the fp32 torch module (the policy rows are exact in both builds)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from alphazero_amd import engine as E                # noqa: E402
from alphazero_amd.models import AlphaZeroC4         # noqa: E402
from alphazero_amd.net import FusedC4NetHX           # noqa: E402
from test_gpu_net import _random_positions, _randomize_bn   # noqa: E402

print('library:', os.environ.get('AZHIP_LIBRARY', 'alphazero_amd/libazhip.so'))
for blocks in (1, 6):
    torch.manual_seed(blocks * 10 + 32)
    model = AlphaZeroC4(blocks, 32).eval()
    _randomize_bn(model, 6)
    hists = _random_positions(256, 12)
    bb, nm = E.positions_from_history(1, hists)
    net = FusedC4NetHX(model, 'cuda')
    x, valid = E.render(1, bb)
    with torch.no_grad():
        _, vref = model(x.cpu(), valid.cpu())
    vref = vref.reshape(-1)
    first, changed, wrong, waves = None, 0, 0, set()
    for run in range(10):
        p, v = net(bb)
        v = v.cpu().clone()
        bad = torch.nonzero((v - vref).abs() > 1e-6).flatten().tolist()
        wrong += len(bad)
        waves |= {i // 32 for i in bad}                  # leaf i is evaluated by wave i // 32 of workgroup i % 32
        if first is None:
            first = v
        else:
            changed += int((v != first).sum())
    print(f'{blocks} blocks: values off by more than 1e-6 in ten runs of 256: {wrong}; values that differ from the first '
          f'run: {changed}; waves of their workgroups that produced a wrong value: {sorted(waves)}')
