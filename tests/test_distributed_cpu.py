"""N > 1 path on CPU: two gloo ranks exercise game-uid sharding, the weight broadcast and the
finished-game gather (the same code runs over RCCL on GPUs)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_records(rank, n):
    from alphazero_amd._lib import GameRecord
    recs = []
    for i in range(n):
        r = GameRecord()
        r.uid = rank * 1000 + i
        r.num_moves = 7 + i
        r.winner = i % 2
        r.model_version = 5
        for m in range(r.num_moves):
            r.history[m] = (m + rank) % 7
            r.num_edges[m] = 7
            for a in range(7):
                r.visits[m][a] = 100 * rank + 10 * m + a
        recs.append(r)
    return recs


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from alphazero_amd import distributed as D
    from alphazero_amd.models import AlphaZeroC4
    # --- sharding: disjoint uid sets
    first, stride = D.shard_uids(rank, world, 8)
    uids = {first + g + k * stride for g in range(8) for k in range(3)}
    gathered = [None] * world
    dist.all_gather_object(gathered, uids)
    assert sum(len(u) for u in gathered) == len(set().union(*gathered))
    # --- weight broadcast: rank 0's weights and version arrive everywhere, aliases intact
    torch.manual_seed(rank)
    model = AlphaZeroC4(2, 8)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.fill_(0.1 * (rank + 1))
    ref = AlphaZeroC4(2, 8)
    torch.manual_seed(0)
    ref0 = AlphaZeroC4(2, 8)
    version = D.broadcast_model(model, 17 if rank == 0 else -1, src=0)
    assert version == 17
    for k, v in model.state_dict().items():
        if 'running_mean' in k:
            assert torch.all(v == 0.1), k
        elif 'num_batches_tracked' not in k and 'running_var' not in k:
            assert torch.equal(v, ref0.state_dict()[k]), k
    sd = model.state_dict()
    assert sd['_tower.0._c1.weight'].data_ptr() == sd['_tower.1._c1.weight'].data_ptr()
    # --- the same through ModelServer.sync (update on the trainer rank, latest everywhere)
    from alphazero_amd.model_server import ModelServer
    srv = ModelServer()
    torch.manual_seed(100 + rank)
    mine = AlphaZeroC4(2, 8)
    if rank == 0:
        srv.update(mine, 9)
    assert srv.sync(mine, src=0) == 9
    fresh = AlphaZeroC4(2, 8)
    assert srv.latest(fresh) == 9
    torch.manual_seed(100)
    want = AlphaZeroC4(2, 8)
    assert all(torch.equal(v, want.state_dict()[k]) for k, v in fresh.state_dict().items())
    # --- record gather: uneven counts, including an empty rank
    recs = _fake_records(rank, 0 if rank == 1 else 3)
    got = D.gather_records(recs, dst=0)
    if rank == 0:
        assert len(got) == 3
        assert sorted(r.uid for r in got) == [0, 1, 2]
        assert got[1].visits[2][3] == 23 and got[2].num_moves == 9
    else:
        assert got == []
    recs = _fake_records(rank, 2 + rank)
    got = D.gather_records(recs, dst=0)
    if rank == 0:
        assert sorted(r.uid for r in got) == [0, 1, 1000, 1001, 1002]
        r = [x for x in got if x.uid == 1002][0]
        assert r.history[3] == 4 and r.visits[5][6] == 156 and r.model_version == 5
    # --- the device-resident forms, on CPU tensors here: records as a fixed-capacity tensor + {count, flags}
    # (caps: the ranks' buffers differ in size in the last case -- engines of different sizes -- and rank 0's is
    # shorter than rank 1's count: it must be padded, not sent short)
    for counts, flag, caps in (((3, 0), 0, (12, 12)), ((2, 5), 2, (12, 12)), ((0, 0), 0, (12, 12)), ((1, 9), 0, (4, 20))):
        n = counts[rank]
        cap = caps[rank]
        buf = torch.zeros(cap, D.RECORD_BYTES, dtype=torch.uint8)
        if n:
            buf[:n] = torch.from_numpy(D.records_to_bytes(_fake_records(rank, n)))
        buf[n:] = 0xEE                                            # stale tail of the ring copy must not travel
        head = torch.tensor([n, flag if rank == 1 else 0], dtype=torch.int32)
        got, flags, mine = D.gather_records_device(buf, head, dst=0)
        assert flags == flag and mine == n
        if rank == 0:
            assert got.shape == (sum(counts), D.RECORD_BYTES)
            uids = [r.uid for r in D.bytes_to_records(got.numpy())]
            assert uids == [i for i in range(counts[0])] + [1000 + i for i in range(counts[1])]
        else:
            assert got.shape[0] == 0
    # --- kernel-ready weights: one in-place broadcast of the evaluator's wire buffer

    class Wire:                                                  # what broadcast_packed needs of a FusedC4Net
        def __init__(self, fill, blocks):
            self.wire = torch.full((256 + 1000,), fill, dtype=torch.uint8)
            self.num_blocks = blocks
    net = Wire(7 if rank == 0 else 9, 6 if rank == 0 else 1)
    where = net.wire.data_ptr()
    assert D.broadcast_packed(net, 41 if rank == 0 else -5, src=0) == 41
    assert net.num_blocks == 6 and net.wire.data_ptr() == where and torch.all(net.wire[256:] == 7)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f'ok{rank}'), 'w').write('ok')


def _dp_worker(rank, world, port, out_dir):
    """Two ranks, each with half of a batch: after allreduce_gradients the SGD step equals the
    single-process step on the whole batch (mean-reduced losses over equal shares), and both ranks
    hold identical weights and batch-norm statistics."""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from alphazero_amd import distributed as D
    from alphazero_amd.loss import AlphaZeroLoss
    from alphazero_amd.models import AlphaZero
    g = torch.Generator().manual_seed(3)
    B = 16
    x = (torch.rand(B, 2, 3, 3, generator=g) < 0.3).float()
    valid = torch.ones(B, 9, dtype=torch.bool)
    p = torch.softmax(torch.randn(B, 9, generator=g), dim=1)
    z = torch.sign(torch.randn(B, 1, generator=g))

    def step(model, sl, sync):
        opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        model.train()
        p_hat, v_hat = model(x[sl], valid[sl])
        pl, vl = AlphaZeroLoss()(p[sl], z[sl], p_hat, v_hat, valid[sl])
        opt.zero_grad()
        (pl + vl).backward()
        if sync:
            D.allreduce_gradients(model)
        opt.step()

    torch.manual_seed(7)
    dp = AlphaZero((2, 3, 3), (9,), 2, 4)
    half = B // world
    step(dp, slice(rank * half, (rank + 1) * half), True)
    torch.manual_seed(7)
    full = AlphaZero((2, 3, 3), (9,), 2, 4)
    step(full, slice(0, B), False)
    for (k, a), (_, b) in zip(dp.named_parameters(), full.named_parameters()):
        # batch-norm normalises each rank's half with that half's statistics, so the two-rank gradient is
        # the full-batch one only up to that difference: close, not equal
        assert torch.allclose(a, b, rtol=0.2, atol=2e-2), k
    flat = torch.cat([t.detach().reshape(-1).float() for t in list(dp.parameters()) + [b for b in dp.buffers() if b.dtype.is_floating_point]])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])                       # the ranks stay bit-identical
    # without batch-norm in the way the equality with the full batch is exact up to summation order
    torch.manual_seed(9)
    lin = torch.nn.Linear(6, 3)
    xs, ys = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    sl = slice(rank * 4, rank * 4 + 4)
    ((lin(xs[sl]) - ys[sl]) ** 2).mean().backward()
    D.allreduce_gradients(lin)
    torch.manual_seed(9)
    lin2 = torch.nn.Linear(6, 3)
    ((lin2(xs) - ys) ** 2).mean().backward()
    assert torch.allclose(lin.weight.grad, lin2.weight.grad, rtol=1e-5, atol=1e-7)
    assert torch.allclose(lin.bias.grad, lin2.bias.grad, rtol=1e-5, atol=1e-7)
    # all-rank record gather (dst=None): every rank receives every record
    recs = D.records_to_bytes(_fake_records(rank, 1 + rank))
    got = D.gather_record_bytes(recs, dst=None)
    assert got.shape[0] == 3 and sorted(r.uid for r in D.bytes_to_records(got)) == [0, 1000, 1001]
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f'dp{rank}'), 'w').write('ok')


def test_two_rank_gloo_data_parallel_step(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'dp0') and os.path.exists(tmp_path / 'dp1')


def test_two_rank_gloo_sharding_broadcast_gather(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no launcher around it must start N ranks itself, through
    torch.distributed.run on 127.0.0.1, passing its own arguments on, before this process has touched a
    GPU (the parent only relays); with WORLD_SIZE set (the launcher's children) it must not."""
    import importlib
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    calls = []
    monkeypatch.setattr(subprocess, 'call', lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: (_ for _ in ()).throw(AssertionError('GPU touched')))
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    else:
        raise AssertionError('the parent must exit with the launcher status')
    (cmd, env), = calls
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '4', '--steps', '3', '--warmup', '1'] and cmd[-7].endswith('bench.py')
    assert env.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'
