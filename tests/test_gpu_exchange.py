"""The hand-over of finished games and the multi-rank exchange on a GPU: device-resident records equal the
host drain byte for byte, engine error bits surface, `bench.py --gpus 2` starts its own two ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sorted_by_uid(raw):
    raw = np.ascontiguousarray(raw)
    return raw[np.argsort(raw[:, :8].copy().view(np.uint64)[:, 0])]


def test_device_hand_over_equals_the_host_drain():
    from alphazero_amd import engine as E
    from alphazero_amd._lib import GameRecord
    G, S = 384, 16
    host, dev = [], []
    for mode in ('host', 'device'):
        eng = E.SelfPlayEngine(E._lib.GAME_C4, G, S, exploration_depth=6, reuse_search_tree=True, seed=5, device=0)
        for mv in range(40):
            eng.begin_move_uniform()
            eng.run_sims_uniform(S)
            eng.choose_and_advance()
            if mv % 3 == 2:
                if mode == 'host':
                    host.append(eng.drain_finished_bytes().copy())
                else:
                    recs, head = eng.take_finished_device()
                    n, flags = head.tolist()
                    assert flags == 0
                    dev.append(recs[:n].cpu().numpy().copy())
        assert eng.error_flags() == 0
        eng.close()
    host, dev = np.concatenate(host), np.concatenate(dev)
    assert host.shape[0] > G and host.shape == dev.shape
    np.testing.assert_array_equal(_sorted_by_uid(host), _sorted_by_uid(dev))
    first = GameRecord.from_buffer_copy(host[0].tobytes())
    assert first.first_model_version == 0 and first.model_version == 0


def test_records_carry_the_first_and_last_model_version():
    from alphazero_amd import engine as E
    from alphazero_amd._lib import GameRecord
    G, S = 64, 8
    eng = E.SelfPlayEngine(E._lib.GAME_C4, G, S, exploration_depth=42, reuse_search_tree=True, seed=9, device=0,
                           model_version=3)
    seen = []
    for mv in range(60):
        if mv == 10:
            eng.set_model_version(4)
        eng.begin_move_uniform()
        eng.run_sims_uniform(S)
        eng.choose_and_advance()
        seen += [(r.first_model_version, r.model_version) for r in eng.drain_finished()]
    eng.close()
    assert (3, 3) in seen or (3, 4) in seen
    assert (3, 4) in seen and (4, 4) in seen            # games that spanned the update, games that started after it
    assert all(a <= b for a, b in seen)


def test_engine_error_bits_surface_at_the_hand_over():
    from alphazero_amd import engine as E
    eng = E.SelfPlayEngine(E._lib.GAME_C4, 4, 64, node_capacity=8, device=0)
    eng.begin_move_uniform()
    eng.run_sims_uniform(64)
    assert eng.error_flags() == 0                         # nothing drained yet
    eng.choose_and_advance()
    eng.drain_finished_bytes()
    assert eng.error_flags() & 1
    recs, head = eng.take_finished_device()
    eng.begin_move_uniform()
    eng.run_sims_uniform(64)
    eng.choose_and_advance()
    recs, head = eng.take_finished_device()
    assert head.tolist()[1] & 1
    eng.reset_games()
    assert eng.error_flags() == 0
    eng.close()


def test_worker_raises_on_pool_exhaustion():
    from alphazero_amd import engine as E
    from alphazero_amd._lib import AzError
    from alphazero_amd.config import AlphaZeroConfig
    from alphazero_amd.selfplay import SelfPlayWorker

    class Server:
        def latest(self, model):
            return 0

    cfg = AlphaZeroConfig(conf={'game': {'module': 'alphazero.games.connectfour', 'name': 'ConnectFour'},
                                'model': {'module': 'alphazero.models', 'name': 'UniformModel'},
                                'selfplay': {'num_expansions_per_sim': 64, 'exploration_depth': 4,
                                             'reuse_search_tree': True}})
    w = SelfPlayWorker(cfg, Server(), None, 'cuda:0', num_games=8)
    w.engine.close()
    w.engine = E.SelfPlayEngine(E._lib.GAME_C4, 8, 64, exploration_depth=4, node_capacity=16, device=0)
    w.refresh_model()
    with pytest.raises(AzError, match='node-pool'):
        for _ in range(3):
            w.play_move(True)
            w.collect_finished_bytes()


def test_bench_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher: both ranks rehearse on this one GPU (gloo), rank 0 reports
    n_gpus 2, the records of both ranks arrive on rank 0."""
    env = dict(os.environ, AZ_BENCH_REHEARSE='1')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--games', '256', '--sims', '32',
           '--mix-moves', '30', '--preroll', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['steps'] == 3 and 'rehearsal' in out
    assert out['value'] > 0 and out['config']['records_gathered_on_rank0'] > 0


def test_bench_exchange_path_over_rccl_with_one_rank():
    """The multi-rank code path of bench.py on the real backend (RCCL: process group with a device id, packed
    weight broadcast, head all_gather, record gather, barrier and max-reduce of the time), forced with a single
    rank because this box has one GPU."""
    env = dict(os.environ, AZ_BENCH_FORCE_DIST='1', MASTER_PORT=str(29600 + os.getpid() % 300))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'AZ_BENCH_REHEARSE'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--games', '512', '--sims', '32',
           '--mix-moves', '30', '--preroll', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert out['n_gpus'] == 1 and out['config']['exchange'].startswith('per step: broadcast_packed')
    assert out['config']['records_gathered_on_rank0'] > 0 and out['value'] > 0
