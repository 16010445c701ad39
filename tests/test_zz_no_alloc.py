"""az_c4_move_fused_static (the entry point that allocates nothing and brings its own staging area)
against az_c4_move_fused, through the C ABI: identical trees, moves, visit counts and records, on two
calls in a row, the second on a smaller engine, so the staging area still holds the first call's
contents when the second runs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _play(G, S, moves, static, net, seed):
    from alphazero_amd import engine as E
    eng = E.SelfPlayEngine(E._lib.GAME_C4, G, S, exploration_depth=4, reuse_search_tree=True, seed=seed, device=0)
    rows = []
    for mv in range(moves):
        bb, info = eng.begin_move()
        p, _ = net(bb, info)
        eng.apply_root(p)
        if static:
            eng.move_fused_static(net)
        else:
            eng.move_fused(net)
        trees = [eng.get_tree(g)[0].copy() for g in range(0, G, max(1, G // 8))]
        a, ne, vis, w = eng.choose_and_advance()
        rows.append((a.cpu().numpy().copy(), vis.cpu().numpy().copy(), w.cpu().numpy().copy(), trees))
    recs = eng.drain_finished_bytes().copy()
    recs = recs[np.argsort(recs[:, :8].copy().view(np.uint64)[:, 0])]
    st = eng.stats()
    assert st['errors'] == 0 and eng.error_flags() == 0
    eng.close()
    return rows, st, recs


def test_static_entry_point_equals_the_workspace_one():
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4Net
    torch.manual_seed(5)
    net = FusedC4Net.from_module(AlphaZeroC4(6, 32).eval().cuda(), 'cuda')
    for G, S, moves, seed in ((600, 40, 9, 31), (96, 25, 12, 32)):       # second call: smaller input
        want = _play(G, S, moves, False, net, seed)
        got = _play(G, S, moves, True, net, seed)
        for (a0, v0, w0, t0), (a1, v1, w1, t1) in zip(want[0], got[0]):
            np.testing.assert_array_equal(a0, a1)
            np.testing.assert_array_equal(v0, v1)
            np.testing.assert_array_equal(w0, w1)
            for x, y in zip(t0, t1):
                np.testing.assert_array_equal(x, y)
        assert want[1] == got[1]
        np.testing.assert_array_equal(want[2], got[2])


def test_static_entry_point_rejects_oversized_engines():
    from alphazero_amd import engine as E
    from alphazero_amd._lib import AzError
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4Net
    torch.manual_seed(5)
    net = FusedC4Net.from_module(AlphaZeroC4(1, 8).eval().cuda(), 'cuda')
    eng = E.SelfPlayEngine(E._lib.GAME_C4, 16385, 2, node_capacity=8, device=0)
    with pytest.raises(AzError, match='16384'):
        eng.move_fused_static(net)
    eng.close()


def test_static_entry_point_allocates_nothing_and_can_be_captured():
    """What the entry point exists for.  (1) Free device memory is the same before and after a call (after one
    warm-up call, so that lazily loaded code objects are in place).  (2) The call can be recorded in a HIP graph
    -- where an allocation, a memset on another stream or a synchronisation would fail the capture -- and the
    replayed launch builds the same trees as a direct az_c4_move_fused on an identical engine."""
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4Net
    torch.manual_seed(6)
    net = FusedC4Net.from_module(AlphaZeroC4(6, 32).eval().cuda(), 'cuda')
    G, S = 300, 30

    def prepared(seed):
        eng = E.SelfPlayEngine(E._lib.GAME_C4, G, S, exploration_depth=4, reuse_search_tree=True, seed=seed, device=0)
        bb, info = eng.begin_move()
        p, _ = net(bb, info)
        eng.apply_root(p)
        return eng

    # (1) no allocation
    eng = prepared(41)
    eng.move_fused_static(net)                                 # warm-up
    eng.choose_and_advance()
    bb, info = eng.begin_move()
    p, _ = net(bb, info)
    eng.apply_root(p)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    eng.move_fused_static(net)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 == free1
    eng.close()

    # (2) under graph capture
    want_eng, got_eng = prepared(42), prepared(42)
    want_eng.move_fused(net)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            got_eng.move_fused_static(net)
    torch.cuda.synchronize()
    assert got_eng.stats()['sims'] == 0                         # recording runs nothing
    graph.replay()
    torch.cuda.synchronize()
    a0, _, v0, _ = want_eng.choose_and_advance()
    a1, _, v1, _ = got_eng.choose_and_advance()
    assert torch.equal(a0, a1) and torch.equal(v0, v1)
    assert v1.sum(1).min().item() >= S
    for e in (want_eng, got_eng):
        e.drain_finished()
        assert e.error_flags() == 0
        e.close()
