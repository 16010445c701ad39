"""The one-launch training pass when a workgroup never arrives: with a -DPASS_TEST_STALL=7 build of the library
(make -C alphazero_amd/csrc OUT=../variants/pass_stall.so EXTRA=-DPASS_TEST_STALL=7, loaded through AZHIP_LIBRARY)
workgroup 1 withholds its row of step 7.  Every reader must give up after about a second, the launch must end, out3
must say so, and the next pass (the error word is cleared by the launch in front of it) must behave the same way.
    AZHIP_LIBRARY=alphazero_amd/variants/pass_stall.so python tools/exp_train_pass_stall.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.train_c4 import C4TrainStep
    torch.manual_seed(0)
    B = 16
    model = AlphaZeroC4(2, 32).cuda().train()
    step = C4TrainStep(model)
    rng = np.random.RandomState(0)
    x = torch.from_numpy((rng.rand(B, 2, 6, 7) < 0.2).astype(np.float32)).cuda()
    x[:, 1] *= 1 - x[:, 0]
    valid = torch.ones(B, 7, dtype=torch.bool, device='cuda')
    p = torch.full((B, 7), 1.0 / 7, device='cuda')
    z = torch.zeros(B, 1, device='cuda')
    for call in range(2):
        t0 = time.perf_counter()
        out3 = step.run(x, p, z, valid)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        a, b, c = (float(v) for v in out3)
        print(f'call {call}: the launch ended after {dt:.2f} s with out3 = {a} {b} {c}')
        assert np.isnan(a) and np.isnan(b) and np.isinf(c), 'the stalled pass must report itself'
        assert dt < 30.0
    # the chain of launches does not depend on the exchange and still works in this build
    chain = C4TrainStep(model, chain=True)
    out3 = chain.run(x, p, z, valid)
    torch.cuda.synchronize()
    assert torch.isfinite(out3).all()
    print('gave up cleanly twice; the chain form is unaffected:', [round(float(v), 6) for v in out3])


if __name__ == '__main__':
    main()
