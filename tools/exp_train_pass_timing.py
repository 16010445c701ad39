"""Where the one-launch training pass spends its time: a -DPASS_TIMING build of the library
(make -C alphazero_amd/csrc OUT=../variants/pass_timing.so EXTRA=-DPASS_TIMING, loaded through AZHIP_LIBRARY) lets
workgroup 0 note the time at which each step's batch sums were complete and at which the step ended.
    AZHIP_LIBRARY=alphazero_amd/variants/pass_timing.so python tools/exp_train_pass_timing.py [batch] [blocks]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.train_c4 import C4TrainStep
    from test_gpu_net import _random_positions
    torch.manual_seed(1)
    model = AlphaZeroC4(blocks, 32).cuda().train()
    rng = np.random.RandomState(3)
    bb, nm = E.positions_from_history(1, _random_positions(B, 9))
    x, valid = E.render(1, bb)
    x, valid = x.contiguous(), valid.contiguous()
    t = rng.random_sample((B, 7)).astype(np.float32) * valid.cpu().numpy()
    t /= t.sum(1, keepdims=True)
    p = torch.from_numpy(t.astype(np.float32)).cuda()
    z = torch.from_numpy(rng.choice([-1.0, 0.0, 1.0], size=(B, 1)).astype(np.float32)).cuda()
    step = C4TrainStep(model)
    for _ in range(20):
        step.run(x, p, z, valid)
    torch.cuda.synchronize()
    st = step._ws[-2048:].view(torch.int64).cpu().numpy().astype(np.int64)
    t0, t1 = st[254], st[255]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); step.run(x, p, z, valid); ev1.record(); torch.cuda.synchronize()
    tick = ev0.elapsed_time(ev1) * 1e3 / (t1 - t0)         # the counter's period from the pass's duration (prep launch included)
    print(f'batch {B}, {blocks} blocks; kernel start to end {(t1 - t0) * tick:.1f} us')
    names = (['stem'] + [f'conv{k}.{j}' for k in range(blocks) for j in (1, 2)] + ['head conv', 'head fc fwd+loss+bwd', 'head conv bwd']
             + [f'dconv{k}.{j}' for k in reversed(range(blocks)) for j in (2, 1)] + ['stem bwd', 'batch sums'])
    prev_end = t0
    for i, name in enumerate(names):
        ready, end = st[2 * i], st[2 * i + 1]
        if i == len(names) - 1:
            end = t1
        wait = (ready - prev_end) * tick if ready else 0.0
        work = (end - (ready if ready else prev_end)) * tick
        print(f'  {name:22s} wait for the batch sums {wait:6.2f} us, rest of the step {work:6.2f} us')
        prev_end = end
    m = st[140:150]
    k = int((m != 0).sum())
    print('  batch sums: from the barrier to the first round %.2f us; rounds' % ((m[0] - st[2 * (len(names) - 1)]) * tick),
          ' '.join('%.2f' % ((m[i + 1] - m[i]) * tick) for i in range(k - 1)), 'us; then to the end %.2f us' % ((t1 - m[k - 1]) * tick))
    d = st[128:133]
    print('  last conv backward step after its sums: weights + coefficients + staging %.2f us, input gradient %.2f us, '
          'mask + sums + post %.2f us, weight gradient %.2f us' % tuple((d[i + 1] - d[i]) * tick for i in range(4)))


if __name__ == '__main__':
    main()

