"""Host time of TrainingWorker.process_batch() (recorded step) by function: cProfile over 300 calls on a filled
device window.   python tools/exp_train_host_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    from alphazero_amd.config import AlphaZeroConfig
    from alphazero_amd.games import ConnectFour
    from alphazero_amd.replay_buffer import DeviceReplayWindow
    from alphazero_amd.training import TrainingWorker
    dev = torch.device('cuda:0')
    tcfg = AlphaZeroConfig(conf={
        'game': {'module': 'alphazero.games.connectfour', 'name': 'ConnectFour'},
        'model': {'module': 'alphazero.games.connectfour', 'name': 'AlphaZeroC4', 'kwargs': {'num_blocks': 6, 'channels_per_block': 32}},
        'loss': {'module': 'alphazero.loss', 'name': 'AlphaZeroLoss'},
        'training': {'num_steps': 10000, 'batch_size': 128, 'lr': 0.2, 'momentum': 0.9, 'weight_decay': 1e-4,
                     'lr_schedule_gamma': 0.1, 'lr_schedule': [1000, 2000, 4000]}})

    class _Publish:
        def update(self, model, version):
            self.version = version
    window = DeviceReplayWindow(ConnectFour, 64 * 1024, dev)
    rng = np.random.RandomState(0)
    n = 49152                                   # (a window this large: one batch in six has two equal first draws)
    x = torch.from_numpy((rng.random_sample((n, 2, 6, 7)) < 0.2).astype(np.float32)).to(dev)
    p = torch.from_numpy(rng.dirichlet(np.ones(7), n).astype(np.float32)).to(dev)
    z = torch.from_numpy(rng.choice([-1.0, 0.0, 1.0], n).astype(np.float32)).to(dev)
    valid = torch.ones(n, 7, dtype=torch.bool, device=dev)
    window.x[:n], window.p[:n], window.z[:n], window.p_valid[:n] = x, p, z.reshape(window.z[:n].shape), valid
    window.count = n
    if hasattr(window, 'count_dev'):
        window.count_dev.fill_(n)
    torch.manual_seed(0)
    tw = TrainingWorker(tcfg, _Publish(), window, None, dev)
    for _ in range(8):
        tw.process_batch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        tw.process_batch()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'300 calls: host {1e6 * (t1 - t0) / 300:.1f} us per call, with the device drained {1e6 * (t2 - t0) / 300:.1f} us per call')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        tw.process_batch()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(18)


if __name__ == '__main__':
    main()
