"""Times the hand-written training pass of the 6 x 32 Connect Four network alone (random batch of 128 boards):
python tools/train_step_probe.py [iterations]; run under rocprofv3 --kernel-trace --stats for the per-kernel view."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.models import AlphaZeroC4          # noqa: E402
from alphazero_amd.optim import FlatSGD               # noqa: E402
from alphazero_amd.train_c4 import C4TrainStep        # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = 128
torch.manual_seed(0)
model = AlphaZeroC4(6, 32).cuda().train()
opt = FlatSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
step = C4TrainStep(model)
rng = np.random.RandomState(0)
x = torch.from_numpy((rng.rand(B, 2, 6, 7) < 0.2).astype(np.float32)).cuda()
x[:, 1] *= 1 - x[:, 0]
valid = torch.ones(B, 7, dtype=torch.bool, device='cuda')
p = torch.full((B, 7), 1.0 / 7, device='cuda')
z = torch.zeros(B, 1, device='cuda')
for _ in range(5):
    step.run(x, p, z, valid)
    opt.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    step.run(x, p, z, valid)
    opt.step()
torch.cuda.synchronize()
print('eager: %.1f us per pass + update' % (1e6 * (time.perf_counter() - t0) / iters))
opt.set_recorded_lr()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step.run(x, p, z, valid)
    opt.step_recorded()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    g.replay()
torch.cuda.synchronize()
print('graph: %.1f us per pass + update' % (1e6 * (time.perf_counter() - t0) / iters))
