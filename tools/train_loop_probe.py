#!/usr/bin/env python3
"""Where a step of the self-play + training loop goes (one GPU): move, record hand-over, window update, SGD step,
re-pack of the weights.  Each part is timed with a device synchronisation behind it, so the parts add up to more
than the loop's own step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_amd import engine as E
from alphazero_amd.config import AlphaZeroConfig
from alphazero_amd.games import ConnectFour
from alphazero_amd.models import AlphaZeroC4
from alphazero_amd.net import FusedC4Net
from alphazero_amd.replay_buffer import DeviceReplayWindow
from alphazero_amd.training import TrainingWorker

G, S = int(os.environ.get('AZ_G', 8192)), int(os.environ.get('AZ_S', 800))
dev = torch.device('cuda:0')
eng = E.SelfPlayEngine(E._lib.GAME_C4, G, S, exploration_depth=4, reuse_search_tree=True, seed=0, device=0)
torch.manual_seed(0)
fused = FusedC4Net.from_module(AlphaZeroC4(6, 32).eval().cuda(), 'cuda')
for i in range(60):
    eng.begin_move_uniform(); eng.run_sims_uniform(S); eng.choose_and_advance()
    if i % 2: eng.drain_finished()
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
T = {}


def part(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t
    return r


def move():
    bb, info = eng.begin_move(); p, _ = fused(bb, info); eng.apply_root(p)
    eng.move_fused(fused); eng.choose_and_advance()


def collect():
    recs, head = eng.take_finished_device()
    n_rec, flags = head.tolist()
    assert flags == 0
    return recs[:n_rec]


while len(window) < 8 * 128:
    move(); window.add_records_device(collect())
tw = TrainingWorker(tcfg, _Publish(), window, None, dev)
for _ in range(3):
    tw.process_batch()
T.clear()
N = 6
for _ in range(N):
    part('move', move)
    recs = part('take finished', collect)
    part('window update', lambda: window.add_records_device(recs))
    if os.environ.get('AZ_SLEEP'):
        time.sleep(float(os.environ['AZ_SLEEP']) * 1e-3)
    if os.environ.get('AZ_TOUCH'):
        part('touch', lambda: (window.x.sum(), tw._optimizer.flat_p.sum()))
    v_new = part('sgd step', tw.process_batch)
    if os.environ.get('AZ_TWICE'):
        part('sgd step again', tw.process_batch)
        t = time.perf_counter(); tw.process_batch(); T['sgd host only'] = T.get('sgd host only', 0.0) + time.perf_counter() - t
    part('re-pack', lambda: fused.load_module(tw.model))
    eng.set_model_version(v_new)
print({k: round(1e3 * v / N, 2) for k, v in T.items()}, 'ms per step; sum', round(1e3 * sum(T.values()) / N, 2))
