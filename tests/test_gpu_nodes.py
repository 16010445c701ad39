"""Node-level numerics on the GPU: PUCT score / arg-max (az_ucb_score), masked soft-max."""
import numpy as np
import pytest
import torch

from helpers import load_json, load_lut, unhex

pytestmark = pytest.mark.gpu


def test_ucb_score_bits_match_reference():
    from alphazero_amd import engine as E
    from alphazero_amd._lib import check, ptr
    cases = load_json('nodes.json')
    # some of the recorded nodes have more visits than the committed table has rows: the table is extended by the
    # same host expression (its first rows are the committed bits)
    reach = max(sum(c['N']) for c in cases) + 1
    lut = E.make_puct_lut(max(reach, load_lut().shape[0]))
    assert lut[:load_lut().shape[0]].tobytes() == load_lut().tobytes()
    eng = E.SelfPlayEngine(0, 1, 1, lut=lut)                 # TTT: 9 edge slots
    n, A = len(cases), 9
    N = np.zeros((n, A), np.uint32); W = np.zeros((n, A), np.float32); P = np.zeros((n, A), np.float32)
    ne = np.zeros(n, np.int32)
    for i, c in enumerate(cases):
        k = c['n']; ne[i] = k
        N[i, :k] = c['N']; W[i, :k] = unhex(c['W'], '<f4'); P[i, :k] = unhex(c['P'], '<f4')
    dN, dW, dP, dne = (torch.from_numpy(a).cuda() for a in (N.view(np.int32), W, P, ne))
    score = torch.zeros(n, A, dtype=torch.float64, device='cuda')
    am = torch.zeros(n, dtype=torch.int32, device='cuda')
    check(eng.lib.az_ucb_score(eng.h, ptr(dN), ptr(dW), ptr(dP), ptr(dne), n, ptr(score), ptr(am), None))
    torch.cuda.synchronize()
    score, am = score.cpu().numpy(), am.cpu().numpy()
    for i, c in enumerate(cases):
        assert score[i, :c['n']].tobytes() == unhex(c['ucb'], '<f8').tobytes(), i
        assert am[i] == c['argmax'], i


def test_masked_softmax_matches_torch():
    from alphazero_amd import engine as E
    rng = np.random.RandomState(0)
    for game, A in ((1, 7), (0, 9)):
        n = 513
        logits = torch.from_numpy(rng.randn(n, A).astype(np.float32) * 3).cuda()
        legal = rng.randint(1, 1 << A, size=n).astype(np.int32)
        info = torch.from_numpy((legal << 16) | 1).cuda()
        p = E.masked_softmax(game, logits, info).cpu()
        for i in range(n):
            m = torch.tensor([(legal[i] >> a) & 1 for a in range(A)], dtype=torch.bool)
            ref = torch.softmax(logits[i].cpu()[m], 0)
            # tolerance: 1e-5 relative, the bound stated for net outputs in fp32
            assert torch.allclose(p[i][m], ref, rtol=1e-5, atol=1e-7)
            assert torch.all(p[i][~m] == 0)


def test_production_lut_equals_the_golden_table_on_this_host():
    """The default PUCT table (engine.make_puct_lut, this host's numpy) has the bits of the committed golden table:
    the one transcendental of the search is the same on the box that runs the benchmark as where the goldens
    were recorded."""
    from alphazero_amd import engine as E
    lut = load_lut()
    assert E.make_puct_lut(lut.shape[0]).tobytes() == lut.tobytes()


def test_visit_sum_beyond_the_table_is_reported_not_computed():
    """The device has no logarithm pinned to numpy's: a search that outgrows its PUCT table keeps going on the
    table's last row and says so (error bit 3), instead of silently leaving the pinned numerics."""
    import torch
    from alphazero_amd import engine as E
    eng = E.SelfPlayEngine(E._lib.GAME_C4, 4, 40, reuse_search_tree=False, lut=E.make_puct_lut(16), device=0)
    eng.begin_move_uniform()
    eng.run_sims_uniform(40)
    eng.choose_and_advance()
    eng.drain_finished()
    assert eng.error_flags() & 8
    eng.close()
    eng = E.SelfPlayEngine(E._lib.GAME_C4, 4, 40, reuse_search_tree=False, device=0)      # default table: long enough
    eng.begin_move_uniform()
    eng.run_sims_uniform(40)
    eng.choose_and_advance()
    eng.drain_finished()
    assert eng.error_flags() == 0
    eng.close()
