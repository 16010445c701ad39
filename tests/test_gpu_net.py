"""Fused HIP network forward against the fp32 torch statement of the same network."""
import numpy as np
import pytest
import torch

import oracle as orc

pytestmark = pytest.mark.gpu


def _random_positions(n, seed):
    rng = np.random.RandomState(seed)
    hists = []
    while len(hists) < n:
        h = []
        depth = rng.randint(0, 31)
        while len(h) < depth and not orc.terminal(orc.C4, h):
            h.append(int(rng.choice(orc.valid_actions(orc.C4, h))))
        if not orc.terminal(orc.C4, h):
            hists.append(h)
    return hists


def _randomize_bn(m, seed):
    g = torch.Generator().manual_seed(seed)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.uniform_(-0.2, 0.2, generator=g)
            mod.running_var.uniform_(0.5, 1.5, generator=g)
            mod.weight.data.uniform_(0.5, 1.5, generator=g)
            mod.bias.data.uniform_(-0.2, 0.2, generator=g)


# the larger sizes give every wave several leaves (tasks of two boards, heads run per four boards,
# a trailing single-board task), the small ones leave most waves idle
@pytest.mark.parametrize('blocks,ch,n', [(6, 32, 1000), (2, 8, 37), (1, 32, 16), (3, 16, 512), (1, 32, 9001),
                                         (2, 16, 6150), (1, 8, 14337)])
def test_fused_bf16_forward_matches_torch_fp32(blocks, ch, n):
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4Net
    torch.manual_seed(blocks * 100 + ch)
    model = AlphaZeroC4(blocks, ch).eval()
    _randomize_bn(model, 5)
    base = _random_positions(min(n, 1200), 11)
    hists = [base[(i * 7) % len(base)] for i in range(n)]
    bb, nm = E.positions_from_history(1, hists)
    legal, _, _ = E.game_query(1, bb, nm)
    info = ((legal << 16) | 1).to(torch.int32)
    net = FusedC4Net.from_module(model, 'cuda')
    p, v = net(bb, info)
    p, v = p.cpu(), v.cpu()
    x, valid = E.render(1, bb)
    with torch.no_grad():
        logits, vref = model(x.cpu(), valid.cpu())
    valid = valid.cpu()
    pref = torch.zeros_like(logits)
    for i in range(n):
        pref[i, valid[i]] = torch.softmax(logits[i, valid[i]], 0)
    assert torch.all(p[~valid] == 0)
    assert torch.allclose(p.sum(1), torch.ones(n), atol=1e-5)
    # bf16 trunk (8-bit mantissa activations and weights, fp32 accumulation): tolerance 5e-3 absolute
    # on probabilities and values, which are O(1) quantities
    assert (p - pref).abs().max().item() < 5e-3, (p - pref).abs().max().item()
    assert (v - vref.reshape(-1)).abs().max().item() < 5e-3, (v - vref.reshape(-1)).abs().max().item()
    print('max |dp|', (p - pref).abs().max().item(), 'max |dv|', (v - vref.reshape(-1)).abs().max().item())


def test_fused_forward_packs_only_leaves_that_need_eval():
    """Leaves whose kind is NONE / TERMINAL are skipped (their output rows are left alone);
    the packed evaluation of the others equals the dense evaluation row for row."""
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4Net
    torch.manual_seed(3)
    model = AlphaZeroC4(2, 32).eval()
    n = 11000
    base = _random_positions(1500, 21)
    hists = [base[(i * 11) % len(base)] for i in range(n)]
    bb, nm = E.positions_from_history(1, hists)
    legal, _, _ = E.game_query(1, bb, nm)
    net = FusedC4Net.from_module(model, 'cuda')
    dense_p, dense_v = net(bb, ((legal << 16) | 1).to(torch.int32))
    dense_p, dense_v = dense_p.clone(), dense_v.clone()
    rng = np.random.RandomState(4)
    for frac in (0.0, 0.03, 0.5, 0.83, 1.0):
        kind = torch.from_numpy(rng.choice([0, 1, 2, 3], size=n, p=[(1 - frac) / 2, frac * 0.9, (1 - frac) / 2, frac * 0.1])
                                .astype(np.int32)).cuda()
        info = ((legal << 16) | kind).to(torch.int32)
        net._p.fill_(-7.0)
        net._v.fill_(-7.0)
        p, v = net(bb, info)
        act = (kind == 1) | (kind == 3)
        assert torch.equal(p[act], dense_p[act]) and torch.equal(v[act], dense_v[act])
        assert torch.all(p[~act] == -7.0) and torch.all(v[~act] == -7.0)


@pytest.mark.parametrize('blocks,ch,n', [(6, 32, 777), (2, 8, 40), (0, 16, 9), (1, 26, 64), (3, 31, 100)])
def test_fused_fp32_forward_matches_torch_fp32(blocks, ch, n):
    """fp32 parity path: net outputs within 1e-5 relative of the fp32 torch statement (p and v are
    O(1); an absolute floor of 1e-6 covers probabilities near zero)."""
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4NetF32
    torch.manual_seed(blocks * 10 + ch)
    model = AlphaZeroC4(blocks, ch).eval()
    _randomize_bn(model, 6)
    hists = _random_positions(n, 12)
    bb, nm = E.positions_from_history(1, hists)
    net = FusedC4NetF32(model, 'cuda')
    p, v = net(bb)
    p, v = p.cpu(), v.cpu()
    x, valid = E.render(1, bb)
    with torch.no_grad():
        logits, vref = model(x.cpu(), valid.cpu())
    valid = valid.cpu()
    pref = torch.zeros_like(logits)
    for i in range(n):
        pref[i, valid[i]] = torch.softmax(logits[i, valid[i]], 0)
    assert torch.all(p[~valid] == 0)
    assert torch.allclose(p, pref, rtol=1e-5, atol=1e-6), (p - pref).abs().max().item()
    assert torch.allclose(v, vref.reshape(-1), rtol=1e-5, atol=1e-6), (v - vref.reshape(-1)).abs().max().item()
    print('fp32 max |dp|', (p - pref).abs().max().item(), 'max |dv|', (v - vref.reshape(-1)).abs().max().item())


@pytest.mark.parametrize('blocks,n', [(2, 301), (10, 130)])
def test_trunk128_bf16_matches_torch_fp32(blocks, n):
    """128-channel trunk (the 10x128 leaf-eval configuration) on bf16 MFMA vs the fp32 torch net."""
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import C4Net128
    torch.manual_seed(40 + blocks)
    model = AlphaZeroC4(blocks, 128).eval()
    _randomize_bn(model, 8)
    with torch.no_grad():               # keep the shared-block tower from amplifying activations
        model._tower[0]._bn2.weight.mul_(0.5)
    hists = _random_positions(n, 13)
    bb, nm = E.positions_from_history(1, hists)
    net = C4Net128(model, 'cuda', max_leaves=512)
    for rep in range(2):                # second call reuses the workspace
        p, v = net(bb)
    p, v = p.cpu(), v.cpu()
    x, valid = E.render(1, bb)
    with torch.no_grad():
        logits, vref = model(x.cpu(), valid.cpu())
    valid = valid.cpu()
    pref = torch.zeros_like(logits)
    for i in range(n):
        pref[i, valid[i]] = torch.softmax(logits[i, valid[i]], 0)
    dp, dv = (p - pref).abs().max().item(), (v - vref.reshape(-1)).abs().max().item()
    print('128ch blocks', blocks, 'max |dp|', dp, 'max |dv|', dv)
    assert torch.all(p[~valid] == 0)
    # bf16 activations and weights through 2*blocks+1 convs of K = 1152: 2e-2 absolute on p, v
    assert dp < 2e-2 and dv < 2e-2


def test_trunk128_full_batch_of_16384_leaves():
    """The 10x128 leaf-evaluation configuration at its full batch (16 384 leaves: 2 731 tower workgroups,
    and the listed path through k_compact128): 512 sampled rows against the fp32 torch net within the
    bf16 tolerance, bit-equal to the same positions evaluated in batches of 130, and bit-equal again when
    the batch is evaluated through a leaf list that skips 15 % of the rows."""
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import C4Net128
    n, blocks = 16384, 10
    torch.manual_seed(50)
    model = AlphaZeroC4(blocks, 128).eval()
    _randomize_bn(model, 8)
    with torch.no_grad():               # keep the shared-block tower from amplifying activations
        model._tower[0]._bn2.weight.mul_(0.5)
    base = _random_positions(1500, 0)
    rng = np.random.RandomState(0)
    pick = rng.randint(0, len(base), size=n)
    hists = [base[i] for i in pick]
    bb, nm = E.positions_from_history(1, hists)
    legal, _, _ = E.game_query(1, bb, nm)
    net = C4Net128(model, 'cuda', max_leaves=n)
    p, v = net(bb)
    p, v = p.clone(), v.clone()
    rows = torch.from_numpy(np.sort(rng.choice(n, size=512, replace=False))).cuda()
    # (a) the same positions in small batches: the rows are pure functions of the position
    small = C4Net128(model, 'cuda', max_leaves=130)
    for lo in range(0, 512, 130):
        idx = rows[lo:lo + 130]
        ps, vs = small(bb[idx].contiguous())
        assert torch.equal(ps, p[idx]) and torch.equal(vs, v[idx])
    # (b) fp32 torch on the sampled rows
    x, valid = E.render(1, bb[rows].contiguous())
    with torch.no_grad():
        logits, vref = model(x.cpu(), valid.cpu())
    valid = valid.cpu()
    pref = torch.zeros_like(logits)
    for i in range(rows.numel()):
        pref[i, valid[i]] = torch.softmax(logits[i, valid[i]], 0)
    dp = (p[rows].cpu() - pref).abs().max().item()
    dv = (v[rows].cpu() - vref.reshape(-1)).abs().max().item()
    print('10x128 at 16384 leaves: max |dp|', dp, 'max |dv|', dv)
    assert torch.all(p[rows].cpu()[~valid] == 0)
    assert dp < 2e-2 and dv < 2e-2
    assert torch.allclose(p.sum(1), torch.ones(n, device='cuda'), atol=1e-4)
    # (c) the listed path: 85 % of the rows need the network, the rest keep their contents
    kind = torch.from_numpy(rng.choice([1, 2, 3, 0], size=n, p=[0.8, 0.1, 0.05, 0.05]).astype(np.int32)).cuda()
    info = ((legal << 16) | kind).to(torch.int32)
    net._p.fill_(-7.0)
    net._v.fill_(-7.0)
    p2, v2 = net(bb, info)
    act = (kind == 1) | (kind == 3)
    assert torch.equal(p2[act], p[act]) and torch.equal(v2[act], v[act])
    assert torch.all(p2[~act] == -7.0) and torch.all(v2[~act] == -7.0)


def test_trunk128_evaluates_only_leaves_that_need_it():
    """With leaf_info the 128-channel path lists the EVAL / ROOT leaves first and runs only those; their
    rows equal the dense evaluation bit for bit, the other rows are left alone."""
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import C4Net128
    torch.manual_seed(9)
    model = AlphaZeroC4(1, 128).eval()
    n = 2500
    base = _random_positions(600, 31)
    hists = [base[(i * 5) % len(base)] for i in range(n)]
    bb, nm = E.positions_from_history(1, hists)
    legal, _, _ = E.game_query(1, bb, nm)
    net = C4Net128(model, 'cuda', max_leaves=n)
    dense_p, dense_v = net(bb)
    dense_p, dense_v = dense_p.clone(), dense_v.clone()
    rng = np.random.RandomState(6)
    for frac in (0.0, 0.01, 0.6, 1.0):
        kind = torch.from_numpy(rng.choice([0, 1, 2, 3], size=n, p=[(1 - frac) / 2, frac * 0.9, (1 - frac) / 2, frac * 0.1])
                                .astype(np.int32)).cuda()
        info = ((legal << 16) | kind).to(torch.int32)
        net._p.fill_(-7.0)
        net._v.fill_(-7.0)
        p, v = net(bb, info)
        act = (kind == 1) | (kind == 3)
        assert torch.equal(p[act], dense_p[act]) and torch.equal(v[act], dense_v[act])
        assert torch.all(p[~act] == -7.0) and torch.all(v[~act] == -7.0)


@pytest.mark.parametrize('n', [1, 2, 3, 5, 255, 2047, 2049, 4097, 16385, 32768])
def test_fused_forward_is_independent_of_batch_shape(n):
    """A leaf's outputs do not depend on how the launch deals leaves to waves: the same positions
    evaluated in one batch of n (dense), with a random subset marked (packed) and in two halves
    give the same bits."""
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4Net
    torch.manual_seed(1)
    model = AlphaZeroC4(1, 8).eval()
    base = _random_positions(min(n, 400), 41)
    hists = [base[(i * 13) % len(base)] for i in range(n)]
    bb, nm = E.positions_from_history(1, hists)
    legal, _, _ = E.game_query(1, bb, nm)
    net = FusedC4Net.from_module(model, 'cuda')
    p0, v0 = net(bb, None)
    p0, v0 = p0.clone(), v0.clone()
    assert torch.allclose(p0.sum(1), torch.ones(n, device='cuda'), atol=1e-5)
    # identical positions give identical rows, wherever they sit in the batch
    if n > len(base):
        j = [i for i in range(1, n) if (i * 13) % len(base) == 0][0]
        assert torch.equal(p0[0], p0[j]) and torch.equal(v0[0], v0[j])
    rng = np.random.RandomState(n)
    kind = torch.from_numpy(rng.choice([0, 1, 2, 3], size=n, p=[0.2, 0.6, 0.1, 0.1]).astype(np.int32)).cuda()
    net._p.fill_(-7.0)
    net._v.fill_(-7.0)
    p1, v1 = net(bb, ((legal << 16) | kind).to(torch.int32))
    act = (kind == 1) | (kind == 3)
    assert torch.equal(p1[act], p0[act]) and torch.equal(v1[act], v0[act])
    assert torch.all(p1[~act] == -7.0) and torch.all(v1[~act] == -7.0)
    if n >= 2:
        h = n // 2
        net2 = FusedC4Net.from_module(model, 'cuda')
        pa, va = net2(bb[:h].contiguous(), None)
        assert torch.equal(pa, p0[:h]) and torch.equal(va, v0[:h])


@pytest.mark.parametrize('blocks,ch,n', [(6, 32, 777), (2, 8, 40), (0, 16, 9), (1, 26, 64), (3, 31, 100), (6, 32, 5000)])
def test_fused_fp16_pair_forward_matches_torch_fp32(blocks, ch, n):
    """The fp32-grade network on the fp16 matrix cores (every operand two fp16 halves, three MFMAs per product,
    csrc/az_net_hx_dev.h) holds the same bar as the fp32 kernel: within 1e-5 relative of the fp32 torch statement,
    absolute floor 1e-6."""
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4NetHX
    torch.manual_seed(blocks * 10 + ch)
    model = AlphaZeroC4(blocks, ch).eval()
    _randomize_bn(model, 6)
    base = _random_positions(min(n, 1200), 12)
    hists = [base[(i * 7) % len(base)] for i in range(n)]
    bb, nm = E.positions_from_history(1, hists)
    net = FusedC4NetHX(model, 'cuda')
    p, v = net(bb)
    p, v = p.cpu(), v.cpu()
    x, valid = E.render(1, bb)
    with torch.no_grad():
        logits, vref = model(x.cpu(), valid.cpu())
    valid = valid.cpu()
    pref = torch.zeros_like(logits)
    for i in range(n):
        pref[i, valid[i]] = torch.softmax(logits[i, valid[i]], 0)
    assert torch.all(p[~valid] == 0)
    assert torch.allclose(p, pref, rtol=1e-5, atol=1e-6), (p - pref).abs().max().item()
    assert torch.allclose(v, vref.reshape(-1), rtol=1e-5, atol=1e-6), (v - vref.reshape(-1)).abs().max().item()
    print('fp16-pair max |dp|', (p - pref).abs().max().item(), 'max |dv|', (v - vref.reshape(-1)).abs().max().item())


@pytest.mark.parametrize('n', [1, 7, 300, 4097])
def test_fused_fp16_pair_forward_is_independent_of_batch_shape(n):
    """Rows are pure functions of the position: dense, packed (leaf_info) and split batches give the same bits,
    and rows of leaves that need no evaluation are left alone."""
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4NetHX
    torch.manual_seed(2)
    model = AlphaZeroC4(2, 32).eval()
    _randomize_bn(model, 8)
    base = _random_positions(min(n, 300), 43)
    hists = [base[(i * 13) % len(base)] for i in range(n)]
    bb, nm = E.positions_from_history(1, hists)
    legal, _, _ = E.game_query(1, bb, nm)
    net = FusedC4NetHX(model, 'cuda')
    p0, v0 = net(bb, None)
    p0, v0 = p0.clone(), v0.clone()
    if n > len(base):
        j = [i for i in range(1, n) if (i * 13) % len(base) == 0][0]
        assert torch.equal(p0[0], p0[j]) and torch.equal(v0[0], v0[j])
    rng = np.random.RandomState(n)
    kind = torch.from_numpy(rng.choice([0, 1, 2, 3], size=n, p=[0.2, 0.6, 0.1, 0.1]).astype(np.int32)).cuda()
    net._p.fill_(-7.0)
    net._v.fill_(-7.0)
    p1, v1 = net(bb, ((legal << 16) | kind).to(torch.int32))
    act = (kind == 1) | (kind == 3)
    assert torch.equal(p1[act], p0[act]) and torch.equal(v1[act], v0[act])
    assert torch.all(p1[~act] == -7.0) and torch.all(v1[~act] == -7.0)
    for i in range(min(n, 5)):
        pa, va = FusedC4NetHX(model, 'cuda')(bb[i:i + 1].contiguous(), None)
        assert torch.equal(pa[0], p0[i]) and torch.equal(va[0], v0[i])


def test_hip_networks_reproduce_the_reference_outputs():
    """The hand-written kernels against the (p, v) the REFERENCE's networks returned while the golden
    games were played (tests/golden/selfplay_net.json; weights rebuilt from the recorded seed):
    fp32 kernel and fp16-pair kernel within 1e-5 relative, bf16 kernel within its 5e-3 tolerance."""
    from helpers import eval_table, golden_net_model, golden_net_positions, load_json
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4Net, FusedC4NetF32, FusedC4NetHX
    checked = 0
    for case in load_json('selfplay_net.json'):
        if case['game'] != 'c4':
            continue
        model = golden_net_model(case, AlphaZeroC4)
        keys, x = golden_net_positions(case, (2, 6, 7))
        tab = eval_table(case)
        w = (1 << (np.arange(7, dtype=np.uint64)[None, :] * np.uint64(7) + np.arange(6, dtype=np.uint64)[:, None]))
        bb_np = np.stack([(x[:, pl].astype(np.uint64) * w[None]).sum(axis=(1, 2)) for pl in (0, 1)], axis=1)
        bb = torch.from_numpy(bb_np.astype(np.int64)).cuda()
        xr, valid = E.render(1, bb)
        assert np.array_equal(xr.cpu().numpy(), x)                   # same positions as the reference rendered
        valid = valid.cpu().numpy().astype(bool)
        for net, rtol, atol in ((FusedC4NetF32(model, "cuda"), 1e-5, 1e-6),
                                (FusedC4NetHX(model, "cuda"), 1e-5, 1e-6),
                                (FusedC4Net.from_module(model, 'cuda'), 0.0, 5e-3)):
            p, v = net(bb, None)
            p, v = p.cpu().numpy(), v.cpu().numpy()
            for i, k in enumerate(keys):
                p_ref, v_ref = tab[k]
                assert np.allclose(p[i][valid[i]], p_ref, rtol=rtol, atol=atol), (case['net_kwargs'], i)
                assert np.all(p[i][~valid[i]] == 0)
                assert abs(v[i] - v_ref) <= atol + rtol * abs(v_ref)
                checked += 1
    assert checked > 3000


def _random_ttt_positions(n, seed):
    rng = np.random.RandomState(seed)
    bb = np.zeros((n, 2), np.int64)
    for i in range(n):
        h = []
        depth = rng.randint(0, 9)
        while len(h) < depth and not orc.terminal(orc.TTT, h):
            h.append(int(rng.choice(orc.valid_actions(orc.TTT, h))))
        if orc.terminal(orc.TTT, h):
            h = h[:-1]
        nx = pv = 0
        for a in h:
            nx, pv = pv, nx | (1 << a)
        bb[i] = (nx, pv)
    return bb


@pytest.mark.parametrize('blocks,ch,n', [(3, 16, 4099), (1, 4, 37), (2, 8, 1000), (0, 16, 5), (3, 16, 1)])
def test_tictactoe_fp32_forward_matches_torch_fp32(blocks, ch, n):
    """az_tttnet_f32_forward against the fp32 torch statement of alphazero/models.py:70-87 on the rendered
    positions, soft-max over the empty squares as alphazero/mcts.py:101-106: rtol 1e-5 / atol 1e-6."""
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZero
    from alphazero_amd.net import TTTNetF32
    torch.manual_seed(blocks * 100 + ch)
    model = AlphaZero((2, 3, 3), (9,), blocks, ch).eval()
    _randomize_bn(model, 5)
    model.cuda()
    bb = torch.from_numpy(_random_ttt_positions(n, n)).cuda()
    x, valid = E.render(0, bb)
    with torch.no_grad():
        logits, v_ref = model(x, valid.bool())
        p_ref = torch.softmax(logits.masked_fill(~valid.bool(), float('-inf')), dim=1)
    net = TTTNetF32(model, 'cuda')
    p, v = net(bb, None)
    assert torch.allclose(p, p_ref, rtol=1e-5, atol=1e-6), float((p - p_ref).abs().max())
    assert torch.allclose(v, v_ref.view(-1), rtol=1e-5, atol=1e-6), float((v - v_ref.view(-1)).abs().max())
    assert torch.all(p[~valid.bool()] == 0)
    # with leaf_info only EVAL / ROOT rows are written; the others keep what was there
    kinds = torch.from_numpy(np.random.RandomState(1).randint(0, 5, size=n).astype(np.int32)).cuda()
    info = kinds | (valid.to(torch.int32) << torch.arange(16, 25, device='cuda', dtype=torch.int32)[None, :]).sum(1).to(torch.int32)
    net._p.fill_(-7.0)
    net._v.fill_(-7.0)
    p2, v2 = net(bb, info)
    need = (kinds == 1) | (kinds == 3)
    assert torch.allclose(p2[need], p_ref[need], rtol=1e-5, atol=1e-6) and torch.all(p2[~need] == -7.0)
    assert torch.allclose(v2[need], v_ref.view(-1)[need], rtol=1e-5, atol=1e-6) and torch.all(v2[~need] == -7.0)


def test_tictactoe_kernel_reproduces_the_reference_outputs():
    """The tic-tac-toe kernel against the (p, v) the REFERENCE's AlphaZero net returned while the golden
    tic-tac-toe game was played (tests/golden/selfplay_net.json): within 1e-5 relative."""
    from helpers import eval_table, golden_net_model, golden_net_positions, load_json
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZero
    from alphazero_amd.net import TTTNetF32
    checked = 0
    for case in load_json('selfplay_net.json'):
        if case['game'] != 'ttt':
            continue
        model = golden_net_model(case, AlphaZero)
        keys, x = golden_net_positions(case, (2, 3, 3))
        tab = eval_table(case)
        w = (1 << np.arange(9, dtype=np.int64)).reshape(3, 3)
        bb_np = np.stack([(x[:, pl].astype(np.int64) * w[None]).sum(axis=(1, 2)) for pl in (0, 1)], axis=1)
        bb = torch.from_numpy(bb_np).cuda()
        xr, valid = E.render(0, bb)
        assert np.array_equal(xr.cpu().numpy(), x)
        valid = valid.cpu().numpy().astype(bool)
        p, v = TTTNetF32(model, 'cuda')(bb, None)
        p, v = p.cpu().numpy(), v.cpu().numpy()
        for i, k in enumerate(keys):
            p_ref, v_ref = tab[k]
            assert np.allclose(p[i][valid[i]], p_ref, rtol=1e-5, atol=1e-6), i
            assert np.all(p[i][~valid[i]] == 0)
            assert abs(v[i] - v_ref) <= 1e-6 + 1e-5 * abs(v_ref)
            checked += 1
    assert checked > 50


@pytest.mark.parametrize('blocks,channels', [(6, 32), (2, 16), (0, 32), (3, 8), (1, 20)])
def test_packing_kernel_writes_the_bytes_of_the_tensor_statement(blocks, channels):
    """az_c4net_pack_bf16 (one launch: batch-norm folded in double, float, bf16, fragment layout) against
    net._pack_c4_blob_torch, the same layout in tensor operations: equal byte for byte, also after the weights and the
    running statistics have moved (FusedC4Net.load_module packs in place)."""
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4Net, _pack_c4_blob_torch, pack_c4_blob
    torch.manual_seed(7 * blocks + channels)
    model = AlphaZeroC4(blocks, channels)
    _randomize_bn(model, blocks + channels)
    with torch.no_grad():
        model._policy_head._fc.bias.uniform_(-0.1, 0.1)
        model._value_head._fc[2].bias.uniform_(-0.1, 0.1)
        model._policy_head._cbr[0].bias.uniform_(-0.2, 0.2)
    model = model.cuda().eval()
    want = _pack_c4_blob_torch(model)
    got = pack_c4_blob(model)
    assert got.dtype == torch.uint8 and got.shape == want.shape
    diff = (got != want).nonzero().reshape(-1)
    assert diff.numel() == 0, (diff.numel(), diff[:8].tolist())
    net = FusedC4Net.from_module(model)
    with torch.no_grad():
        for q in model.parameters():
            q.add_(0.01 * torch.randn_like(q))
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.add_(0.05)
                m.running_var.mul_(1.3)
    net.load_module(model)
    assert torch.equal(net.blob, _pack_c4_blob_torch(model))


@pytest.mark.parametrize('blocks,channels', [(6, 32), (2, 16), (0, 32), (3, 8)])
def test_fp16_pair_packing_kernel_writes_the_bytes_of_the_tensor_statement(blocks, channels):
    """az_c4net_pack_hx against net._pack_c4_blob_hx_torch: equal byte for byte; a weight beyond the fp16 range is
    reported (raised by pack_c4_blob_hx, a sticky device flag for FusedC4NetHX.load_module)."""
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4NetHX, _pack_c4_blob_hx_torch, pack_c4_blob_hx
    torch.manual_seed(11 * blocks + channels)
    model = AlphaZeroC4(blocks, channels)
    _randomize_bn(model, blocks + channels + 1)
    with torch.no_grad():
        model._policy_head._fc.bias.uniform_(-0.1, 0.1)
        model._value_head._fc[2].bias.uniform_(-0.1, 0.1)
        model._policy_head._cbr[0].bias.uniform_(-0.2, 0.2)
    model = model.cuda().eval()
    want = _pack_c4_blob_hx_torch(model)
    got = pack_c4_blob_hx(model)
    diff = (got != want).nonzero().reshape(-1)
    assert got.shape == want.shape and diff.numel() == 0, (diff.numel(), diff[:8].tolist())
    net = FusedC4NetHX(model)
    with torch.no_grad():
        for q in model.parameters():
            q.add_(0.01 * torch.randn_like(q))
    net.load_module(model, check=True)
    assert torch.equal(net.blob, _pack_c4_blob_hx_torch(model)) and int(net.range_error) == 0
    with torch.no_grad():
        model._cbr[0].weight[0, 0, 0, 0] = 1.0e6
    with pytest.raises(ValueError):
        pack_c4_blob_hx(model)
    net.load_module(model)
    assert int(net.range_error) == 1
    with pytest.raises(ValueError):
        net.load_module(model, check=True)
