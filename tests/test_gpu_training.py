"""The training step on the GPU (alphazero/training.py:42-75): the hand-written loss and optimiser kernels
against their torch statements, and four whole steps against the scalars the reference's own TrainingWorker
logged for the same seeds, data and configuration."""
import numpy as np
import pytest
import torch

from helpers import load_json, unhex

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('A,B', [(7, 128), (9, 37), (7, 1), (9, 1000)])
def test_loss_kernel_matches_the_torch_statement(A, B):
    from alphazero_amd.loss import AlphaZeroLoss, masked_cross_entropy
    g = torch.Generator().manual_seed(A * 1000 + B)
    valid = torch.rand(B, A, generator=g) < 0.7
    valid[:, 0] |= ~valid.any(1)                                   # at least one legal move
    t = torch.rand(B, A, generator=g) * valid
    t[torch.rand(B, A, generator=g) < 0.2] = 0                     # zero targets inside the mask
    t[:, 0] += (t.sum(1) == 0) * valid[:, 0]
    t = t * valid
    fix = t.sum(1) == 0
    t[fix] = valid[fix].float()
    t = t / t.sum(1, keepdim=True)
    z = torch.sign(torch.randn(B, 1, generator=g))
    logits = (3 * torch.randn(B, A, generator=g)).requires_grad_()
    v = torch.tanh(torch.randn(B, 1, generator=g)).requires_grad_()
    # torch statement on the CPU (the one pinned to the reference's recorded scalars)
    pl = masked_cross_entropy(logits, t, valid).mean()
    vl = ((z - v) ** 2).mean()
    (1.7 * pl + 0.3 * vl).backward()
    lg, vg = logits.detach().cuda().requires_grad_(), v.detach().cuda().requires_grad_()
    pl2, vl2 = AlphaZeroLoss()(t.cuda(), z.cuda(), lg, vg, valid.cuda())
    (1.7 * pl2 + 0.3 * vl2).backward()
    assert abs(pl2.item() - pl.item()) <= 1e-6 * max(1, abs(pl.item()))
    assert abs(vl2.item() - vl.item()) <= 1e-6 * max(1, abs(vl.item()))
    assert torch.allclose(lg.grad.cpu(), logits.grad, rtol=1e-5, atol=1e-7)
    assert torch.allclose(vg.grad.cpu(), v.grad, rtol=1e-5, atol=1e-7)
    assert torch.all(lg.grad.cpu()[~valid] == 0)


def test_loss_kernel_keeps_the_reference_assertion():
    from alphazero_amd.loss import AlphaZeroLoss
    p = torch.full((4, 7), 0.2, device='cuda')                     # sums to 1.4
    with pytest.raises(AssertionError, match='valid distribution'):
        AlphaZeroLoss()(p, torch.zeros(4, 1, device='cuda'), torch.zeros(4, 7, device='cuda'),
                        torch.zeros(4, 1, device='cuda'), torch.ones(4, 7, dtype=torch.bool, device='cuda'))


def test_flat_sgd_matches_torch_sgd():
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.optim import FlatSGD
    from torch.optim.lr_scheduler import MultiStepLR
    torch.manual_seed(4)
    a = AlphaZeroC4(2, 8).cuda()
    torch.manual_seed(4)
    b = AlphaZeroC4(2, 8).cuda()
    oa = torch.optim.SGD(a.parameters(), lr=0.2, momentum=0.9, weight_decay=1e-4)
    ob = FlatSGD(b.parameters(), lr=0.2, momentum=0.9, weight_decay=1e-4)
    sa, sb = MultiStepLR(oa, [2, 4], gamma=0.1), MultiStepLR(ob, [2, 4], gamma=0.1)
    sd = b.state_dict()
    assert sd['_tower.0._c1.weight'].data_ptr() == sd['_tower.1._c1.weight'].data_ptr()     # aliases survive the flattening
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        x = (torch.rand(32, 2, 6, 7, generator=g) < 0.3).float().cuda()
        valid = torch.ones(32, 7, dtype=torch.bool, device='cuda')
        for m, o, s in ((a, oa, sa), (b, ob, sb)):
            m.train()
            o.zero_grad()
            p_hat, v_hat = m(x, valid)
            (p_hat.square().mean() + v_hat.square().mean()).backward()
            o.step()
            s.step()
        assert sa.get_last_lr() == sb.get_last_lr()
        for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            assert torch.allclose(pa, pb, rtol=2e-5, atol=1e-6), (step, k, (pa - pb).abs().max().item())


def test_training_steps_on_the_gpu_match_the_reference_scalars():
    """host.json 'training': four process_batch() calls of the reference's TrainingWorker on the CPU.  The same
    seeds, games and configuration here with the model on the GPU, the HIP loss kernel and FlatSGD: every logged
    scalar within 1e-5 relative, the final weights within 1e-4 (vendor convolutions, different summation order)."""
    from test_host_cpu import _golden_game
    from alphazero_amd.config import AlphaZeroConfig
    from alphazero_amd.optim import FlatSGD
    from alphazero_amd.replay_buffer import ReplayBufferDataset
    from alphazero_amd.training import TrainingWorker
    t = load_json('host.json')['training']
    cfg = AlphaZeroConfig(conf=t['conf'])
    ds = ReplayBufferDataset(cfg)
    for g in t['games']:
        ds.add_game(_golden_game(cfg, g), 1)

    class Server:
        versions = []

        def update(self, model, version):
            self.versions.append(int(version))

    class Writer:
        rows = []

        def add_scalar(self, tag, value, step):
            self.rows.append([tag, float(value), int(step)])

    torch.manual_seed(t['model_seed'])
    np.random.seed(t['sampler_seed'])
    server, writer = Server(), Writer()
    tw = TrainingWorker(cfg, server, ds, writer, 'cuda', 0)
    assert isinstance(tw._optimizer, FlatSGD)
    assert [tw.process_batch() for _ in range(4)] == t['returns'] and server.versions == t['versions']
    assert [(r[0], r[2]) for r in writer.rows] == [(r[0], r[2]) for r in t['scalars']]
    worst = 0.0
    for got, want in zip(writer.rows, t['scalars']):
        err = abs(got[1] - want[1]) / max(1.0, abs(want[1]))
        worst = max(worst, err)
        assert err <= 1e-5, (got, want)
    state = tw.model.state_dict()
    for k, h in t['final'].items():
        a, b = state[k].detach().cpu().numpy().reshape(-1), unhex(h, '<f4')
        assert np.allclose(a, b, rtol=1e-4, atol=1e-5), (k, np.abs(a - b).max())
    print('worst relative scalar error vs the reference:', worst)


@pytest.mark.parametrize('blocks,channels,hand', [(2, 16, False), (3, 32, True)])
def test_recorded_training_step_equals_the_eager_one(blocks, channels, hand):
    """(Both forms of the pass: PyTorch's forward / backward for a 16-channel network, the hand-written kernels
    of csrc/az_train_c4.hip for a 32-channel one.)  TrainingWorker on a DeviceReplayWindow records its step as a HIP graph after three eager steps
    (training.TrainingWorker._record_step).  Two workers from the same seeds, one recording and one not, must draw the
    same batches and end with the same weights, batch-norm statistics and logged losses, across a change of the
    learning rate by the schedule (the recorded update reads the rate from device memory)."""
    from alphazero_amd import engine as E
    from alphazero_amd.config import AlphaZeroConfig
    from alphazero_amd.games import ConnectFour
    from alphazero_amd.replay_buffer import DeviceReplayWindow
    from alphazero_amd.training import TrainingWorker
    cfg = AlphaZeroConfig(conf={
        'game': {'module': 'alphazero.games.connectfour', 'name': 'ConnectFour'},
        'model': {'module': 'alphazero.games.connectfour', 'name': 'AlphaZeroC4',
                  'kwargs': {'num_blocks': blocks, 'channels_per_block': channels}},
        'loss': {'module': 'alphazero.loss', 'name': 'AlphaZeroLoss'},
        'training': {'num_steps': 100, 'batch_size': 32, 'lr': 0.1, 'momentum': 0.9, 'weight_decay': 1e-4,
                     'lr_schedule_gamma': 0.1, 'lr_schedule': [6]}})
    dev = torch.device('cuda:0')
    eng = E.SelfPlayEngine(E._lib.GAME_C4, 256, 16, exploration_depth=4, reuse_search_tree=True, seed=5, device=0)
    window = DeviceReplayWindow(ConnectFour, 4096, dev)
    while len(window) < 512:
        eng.begin_move_uniform(); eng.run_sims_uniform(16); eng.choose_and_advance()
        window.add_records(eng.drain_finished_bytes())

    class Server:
        def update(self, model, version):
            pass

    class Writer:
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, value, step):
            self.rows.append((tag, float(value), int(step)))

    runs = []
    for record in (False, True):
        torch.manual_seed(11)
        writer = Writer()
        tw = TrainingWorker(cfg, Server(), window, writer, dev, record_step=record)
        assert tw._record == record and (tw._hand is not None) == hand
        assert [tw.process_batch() for _ in range(10)] == list(range(1, 11))
        assert (tw._graph is not None) == record
        runs.append((writer.rows, {k: v.detach().clone() for k, v in tw.model.state_dict().items()}))
    (rows_a, sd_a), (rows_b, sd_b) = runs
    assert [(r[0], r[2]) for r in rows_a] == [(r[0], r[2]) for r in rows_b]
    for ra, rb in zip(rows_a, rows_b):
        # (1e-5: with PyTorch's forward / backward the two runs have been seen 3e-6 apart at the seventh step)
        assert abs(ra[1] - rb[1]) <= 1e-5 * max(1.0, abs(ra[1])), (ra, rb)
    assert any(r[0] == 'optim/learning_rate' and abs(r[1] - 0.01) < 1e-9 for r in rows_b)      # the schedule stepped
    for k in sd_a:
        assert torch.allclose(sd_a[k].float(), sd_b[k].float(), rtol=1e-4, atol=1e-5), (k, (sd_a[k].float() - sd_b[k].float()).abs().max())


@pytest.mark.parametrize('channels', [8, 32])
def test_recorded_step_withholds_the_update_of_a_bad_batch(channels):
    """The reference asserts 'Target policy must be a valid distribution' before optimizer.step()
    (alphazero/loss.py:58-59, training.py:59-71): a bad batch never changes a weight.  The recorded step cannot
    read the check back before its own update, so the update kernel is guarded on the device
    (az_sgd_momentum_guarded): with corrupted targets in the window nothing changes and the withheld counter runs,
    and the host's periodic read-back raises."""
    from alphazero_amd import engine as E
    from alphazero_amd.config import AlphaZeroConfig
    from alphazero_amd.games import ConnectFour
    from alphazero_amd.replay_buffer import DeviceReplayWindow
    from alphazero_amd.training import TrainingWorker
    cfg = AlphaZeroConfig(conf={
        'game': {'module': 'alphazero.games.connectfour', 'name': 'ConnectFour'},
        'model': {'module': 'alphazero.games.connectfour', 'name': 'AlphaZeroC4',
                  'kwargs': {'num_blocks': 1, 'channels_per_block': channels}},
        'loss': {'module': 'alphazero.loss', 'name': 'AlphaZeroLoss'},
        'training': {'num_steps': 100, 'batch_size': 16, 'lr': 0.1, 'momentum': 0.9, 'weight_decay': 1e-4,
                     'lr_schedule_gamma': 0.1, 'lr_schedule': [50]}})
    dev = torch.device('cuda:0')
    eng = E.SelfPlayEngine(E._lib.GAME_C4, 128, 8, exploration_depth=4, reuse_search_tree=True, seed=9, device=0)
    window = DeviceReplayWindow(ConnectFour, 2048, dev)
    while len(window) < 256:
        eng.begin_move_uniform(); eng.run_sims_uniform(8); eng.choose_and_advance()
        window.add_records(eng.drain_finished_bytes())

    class Server:
        def update(self, model, version):
            pass
    torch.manual_seed(3)
    tw = TrainingWorker(cfg, Server(), window, None, dev, record_step=True)
    tw.CHECK_EVERY = 4
    for _ in range(5):                                         # three eager steps, then the recorded one twice
        tw.process_batch()
    assert tw._graph is not None and int(tw._optimizer.withheld) == 0
    before = tw._optimizer.flat_p.clone()
    window.p.mul_(1.5)                                         # every target row now sums to 1.5
    raised = False
    for _ in range(4):
        try:
            tw.process_batch()
        except AssertionError as e:
            assert 'valid distribution' in str(e)
            raised = True
            break
    assert raised
    assert int(tw._optimizer.withheld) >= 1
    assert torch.equal(tw._optimizer.flat_p, before)           # not one weight moved
    window.p.div_(1.5)

    # the guarded kernel against the plain one on a passing check
    from alphazero_amd.optim import FlatSGD
    torch.manual_seed(4)
    a = [torch.nn.Parameter(torch.randn(300, device=dev)), torch.nn.Parameter(torch.randn(7, 5, device=dev))]
    b = [torch.nn.Parameter(t.detach().clone()) for t in a]
    oa, ob = FlatSGD(a, 0.05, 0.9, 1e-4), FlatSGD(b, 0.05, 0.9, 1e-4)
    ok = torch.zeros(1, device=dev)
    for step in range(3):
        g = torch.randn(oa.flat_g.numel(), device=dev)
        oa.flat_g.copy_(g); ob.flat_g.copy_(g)
        oa.set_recorded_lr(); ob.set_recorded_lr()
        oa.step_recorded()
        ob.step_recorded(guard=ok)
    assert torch.equal(oa.flat_p, ob.flat_p) and int(ob.withheld) == 0


@pytest.mark.parametrize('blocks,B,chain', [(6, 128, False), (6, 128, True), (1, 16, False), (0, 8, False), (2, 77, True),
                                             (2, 300, False)])
def test_hand_written_training_pass_matches_torch_autograd(blocks, B, chain):
    """az_c4_train_fwd_bwd (forward in training mode, AlphaZeroLoss, backward: no vendor convolution, no autograd; one
    launch for the whole pass, or -- chain, and any batch beyond one board per compute unit -- one launch per step)
    against the torch statement of alphazero/training.py:57-71 evaluated in float64 on the same GPU: losses and
    network outputs within 1e-5, running statistics and counters as torch leaves them, every gradient within 1e-4 of
    its tensor's scale on the first pass and within 5e-3 on the two passes after an SGD step: there a float32
    evaluation flips the ReLU of a few activations that sit at zero, and the float32 torch module (printed beside it)
    is itself up to 8e-4 away from its float64 twin."""
    import copy
    from alphazero_amd import engine as E
    from alphazero_amd.loss import masked_cross_entropy
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.train_c4 import C4TrainStep, supports
    from test_gpu_net import _random_positions, _randomize_bn
    torch.manual_seed(100 + blocks)
    model = AlphaZeroC4(blocks, 32)
    _randomize_bn(model, 3)
    with torch.no_grad():                      # zero-initialised biases would hide an error in how they are used
        model._policy_head._fc.bias.uniform_(-0.1, 0.1)
        model._value_head._fc[2].bias.uniform_(-0.1, 0.1)
    model = model.cuda().train()
    ref32, ref = copy.deepcopy(model), copy.deepcopy(model).double()
    assert supports(model)
    step = C4TrainStep(model, chain=chain)
    rng = np.random.RandomState(blocks * 7 + B)
    for it in range(3):
        hists = _random_positions(B, 50 + it)
        bb, nm = E.positions_from_history(1, hists)
        x, valid = E.render(1, bb)
        x, valid = x.contiguous(), valid.contiguous()
        t = rng.random_sample((B, 7)).astype(np.float32) * valid.cpu().numpy()
        t /= t.sum(1, keepdims=True)
        p = torch.from_numpy(t.astype(np.float32)).cuda()
        z = torch.from_numpy(rng.choice([-1.0, 0.0, 1.0], size=(B, 1)).astype(np.float32)).cuda()
        out3 = step.run(x, p, z, valid)
        for m_, dt in ((ref32, torch.float32), (ref, torch.float64)):
            for q in m_.parameters():
                q.grad = None
            p_hat, v_hat = m_(x.to(dt), valid)
            policy = masked_cross_entropy(p_hat, p.to(dt), valid).mean()
            value = ((z.to(dt) - v_hat) ** 2).mean()
            (policy + value).backward()
        torch.cuda.synchronize()
        assert abs(out3[0].item() - policy.item()) <= 1e-5 * max(1.0, abs(policy.item())), (out3[0].item(), policy.item())
        assert abs(out3[1].item() - value.item()) <= 1e-5 * max(1.0, abs(value.item())), (out3[1].item(), value.item())
        assert out3[2].item() < 1e-6
        assert torch.allclose(step.logits.double(), p_hat, rtol=1e-5, atol=2e-6), (step.logits - p_hat).abs().max().item()
        assert torch.allclose(step.v_hat.double(), v_hat.reshape(-1), rtol=1e-5, atol=2e-6)
        worst = worst32 = 0.0
        for (name, a), (_, b), (_, c) in zip(model.named_parameters(), ref.named_parameters(), ref32.named_parameters()):
            if name == '_policy_head._cbr.0.bias':
                # this bias feeds a batch-norm: its gradient is zero in exact arithmetic, rounding noise in any other
                assert a.grad.abs().max().item() < 1e-6 and b.grad.abs().max().item() < 1e-6
                continue
            scale = max(b.grad.abs().max().item(), 1e-4)
            err = (a.grad.double() - b.grad).abs().max().item() / scale
            worst = max(worst, err)
            worst32 = max(worst32, (c.grad.double() - b.grad).abs().max().item() / scale)
            assert err < (1e-4 if it == 0 else 5e-3), (it, name, err, scale)
        for (name, a), (_, b) in zip(model.named_buffers(), ref.named_buffers()):
            if a.dtype == torch.int64:
                assert torch.equal(a, b), name
            else:
                assert torch.allclose(a.double(), b, rtol=1e-5, atol=1e-6), (it, name, (a - b).abs().max().item())
        print(f'blocks {blocks} batch {B} pass {it}: worst gradient error / scale {worst:.2e} (torch float32: {worst32:.2e})')
        # all three modules take the same plain SGD step, so that the next pass starts from changed weights
        with torch.no_grad():
            for a, b, c in zip(model.parameters(), ref.parameters(), ref32.parameters()):
                g = b.grad.float()
                a.add_(g, alpha=-0.05)
                c.add_(g, alpha=-0.05)
                b.copy_(a.double())


def _c4_batch(B, seed, rng):
    from alphazero_amd import engine as E
    from test_gpu_net import _random_positions
    bb, nm = E.positions_from_history(1, _random_positions(B, seed))
    x, valid = E.render(1, bb)
    x, valid = x.contiguous(), valid.contiguous()
    t = rng.random_sample((B, 7)).astype(np.float32) * valid.cpu().numpy()
    t /= t.sum(1, keepdims=True)
    p = torch.from_numpy(t.astype(np.float32)).cuda()
    z = torch.from_numpy(rng.choice([-1.0, 0.0, 1.0], size=(B, 1)).astype(np.float32)).cuda()
    return x, p, z, valid


@pytest.mark.parametrize('blocks,B', [(6, 128), (1, 16), (0, 8), (2, 77), (3, 256), (6, 2), (14, 40)])
def test_one_launch_training_pass_equals_the_chain_of_launches(blocks, B):
    """The two forms of az_c4_train_fwd_bwd run the same steps with the same thread-to-data mapping and differ only in
    how the boards' partial sums travel between steps (tagged pairs polled inside one resident grid / rows read after
    a launch boundary): losses, outputs, every gradient and every batch-norm buffer are equal bit for bit, over three
    passes with an SGD step in between (a stale row of the previous pass or step would show)."""
    import copy
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.train_c4 import C4TrainStep
    from test_gpu_net import _randomize_bn
    torch.manual_seed(200 + blocks)
    model = AlphaZeroC4(blocks, 32)
    _randomize_bn(model, 3)
    with torch.no_grad():
        model._policy_head._fc.bias.uniform_(-0.1, 0.1)
        model._value_head._fc[2].bias.uniform_(-0.1, 0.1)
    one = model.cuda().train()
    chain = copy.deepcopy(one)
    s_one, s_chain = C4TrainStep(one), C4TrainStep(chain, chain=True)
    assert s_one.one_launch_max_batch() >= B
    rng = np.random.RandomState(blocks * 7 + B)
    for it in range(3):
        x, p, z, valid = _c4_batch(B, 60 + it, rng)
        a = s_one.run(x, p, z, valid).clone()
        b = s_chain.run(x, p, z, valid).clone()
        torch.cuda.synchronize()
        assert torch.isfinite(a).all(), a
        assert torch.equal(a, b), (it, a, b)
        assert torch.equal(s_one.logits, s_chain.logits) and torch.equal(s_one.v_hat, s_chain.v_hat)
        for (name, u), (_, v) in zip(one.named_parameters(), chain.named_parameters()):
            assert torch.equal(u.grad, v.grad), (it, name, (u.grad - v.grad).abs().max().item())
        for (name, u), (_, v) in zip(one.named_buffers(), chain.named_buffers()):
            assert torch.equal(u, v), (it, name)
        with torch.no_grad():
            for u, v in zip(one.parameters(), chain.parameters()):
                u.add_(u.grad, alpha=-0.05)
                v.add_(v.grad, alpha=-0.05)


def test_one_launch_training_pass_over_many_launches_and_the_wrap_of_its_launch_counter():
    """The rows that the workgroups of the one-launch pass exchange are recognised by a tag made of the launch counter
    (20 bits, kept in the workspace) and the step.  3 000 passes in a row on one workspace never give up and keep
    giving the chain's bits; so do the passes around the counter's wrap (the counter is put just below it by hand)."""
    import copy
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.train_c4 import C4TrainStep
    torch.manual_seed(5)
    B = 16
    one = AlphaZeroC4(1, 32).cuda().train()
    chain = copy.deepcopy(one)
    s_one, s_chain = C4TrainStep(one), C4TrainStep(chain, chain=True)
    rng = np.random.RandomState(1)
    x, p, z, valid = _c4_batch(B, 3, rng)
    want = s_chain.run(x, p, z, valid).clone()
    grads = [q.grad.clone() for q in chain.parameters()]
    for it in range(3000):
        got = s_one.run(x, p, z, valid)
        if it % 500 == 499:
            assert torch.equal(got, want), (it, got, want)
    # int32 {launches so far, error} sits in front of the last two areas of the workspace: 64 floats of its own, then
    # 4 floats per board (rounded up to 64) and 512 floats of time stamps
    ctl = s_one._ws[-4 * (512 + 64 + 64):][:8].view(torch.int32)
    assert int(ctl[0]) == 3000 and int(ctl[1]) == 0
    ctl[0] = 0xFFFFF - 20
    for it in range(40):
        got = s_one.run(x, p, z, valid)
        assert torch.equal(got, want), (it, got, want)
    assert int(ctl[0]) == 0xFFFFF + 20
    for a, b in zip(one.parameters(), grads):
        assert torch.equal(a.grad, b)
    # the workspace needs no initialisation: whatever it holds (stale rows, a set error word), a pass gives the same
    for it in range(3):
        s_one._ws.random_(0, 256)
        got = s_one.run(x, p, z, valid)
        assert torch.equal(got, want), (it, got, want)
    for a, b in zip(one.parameters(), grads):
        assert torch.equal(a.grad, b)


def test_one_launch_training_pass_gives_up_when_a_workgroup_does_not_arrive():
    """The resident grid of the one-launch pass must drain whatever happens.  A test build of the library
    (-DPASS_TEST_STALL=7, made by __graft_entry__.build()) lets workgroup 1 withhold its row of step 7; in a child
    process that loads it, every reader gives up after a fraction of a second, the launch ends with out3 = {NaN, NaN,
    inf}, a second pass does the same (the error word is cleared in between) and the chain form still works."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variant = os.path.join(root, 'alphazero_amd', 'variants', 'pass_stall.so')
    if not os.path.exists(variant):
        pytest.skip('alphazero_amd/variants/pass_stall.so is not built (__graft_entry__.build() makes it)')
    env = dict(os.environ, AZHIP_LIBRARY=variant)
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'exp_train_pass_stall.py')], env=env, capture_output=True,
                       text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'gave up cleanly twice' in r.stdout, r.stdout


def test_training_worker_with_the_hand_written_pass_follows_the_torch_path(monkeypatch):
    """Six process_batch() calls of TrainingWorker on the shipped 6 x 32 network (config/connectfour.yaml:8-9) with
    the hand-written pass and with PyTorch's forward / backward (AZ_TRAIN_HANDWRITTEN=0) from the same seeds and
    window: same batches; after the first step (from identical weights) weights and batch-norm statistics within 2e-4 of
    their tensor's scale and the logged losses within 1e-5; after that the two float32 evaluations drift apart as
    any two do (a ReLU that flips for one of them, then 1e-3 within two steps: see the float64 comparison above), so
    the later steps are only held to 2e-2 of the tensor's scale (+ 1e-4 for the tensors that are themselves 1e-3) and
    2e-3 for the losses: enough to catch a wrong schedule, momentum or batch, which would show at order one."""
    from alphazero_amd import engine as E
    from alphazero_amd.config import AlphaZeroConfig
    from alphazero_amd.games import ConnectFour
    from alphazero_amd.replay_buffer import DeviceReplayWindow
    from alphazero_amd.training import TrainingWorker
    cfg = AlphaZeroConfig(conf={
        'game': {'module': 'alphazero.games.connectfour', 'name': 'ConnectFour'},
        'model': {'module': 'alphazero.games.connectfour', 'name': 'AlphaZeroC4',
                  'kwargs': {'num_blocks': 6, 'channels_per_block': 32}},
        'loss': {'module': 'alphazero.loss', 'name': 'AlphaZeroLoss'},
        'training': {'num_steps': 100, 'batch_size': 128, 'lr': 0.02, 'momentum': 0.9, 'weight_decay': 1e-4,
                     'lr_schedule_gamma': 0.1, 'lr_schedule': [50]}})
    dev = torch.device('cuda:0')
    eng = E.SelfPlayEngine(E._lib.GAME_C4, 256, 16, exploration_depth=4, reuse_search_tree=True, seed=6, device=0)
    window = DeviceReplayWindow(ConnectFour, 4096, dev)
    while len(window) < 1024:
        eng.begin_move_uniform(); eng.run_sims_uniform(16); eng.choose_and_advance()
        window.add_records(eng.drain_finished_bytes())

    class Server:
        def update(self, model, version):
            pass

    class Writer:
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, value, step):
            self.rows.append((tag, float(value), int(step)))
    runs = []
    for hand in ('1', '0'):
        monkeypatch.setenv('AZ_TRAIN_HANDWRITTEN', hand)
        torch.manual_seed(21)
        writer = Writer()
        tw = TrainingWorker(cfg, Server(), window, writer, dev, record_step=False)
        assert (tw._hand is not None) == (hand == '1')
        snaps = []
        for _ in range(6):
            tw.process_batch()
            snaps.append({k: v.detach().clone().float() for k, v in tw.model.state_dict().items()})
        runs.append((writer.rows, snaps))
    (rows_a, snaps_a), (rows_b, snaps_b) = runs
    assert [(r[0], r[2]) for r in rows_a] == [(r[0], r[2]) for r in rows_b]
    for ra, rb in zip(rows_a, rows_b):
        assert abs(ra[1] - rb[1]) <= (1e-5 if ra[2] < 1 else 2e-3) * max(1.0, abs(ra[1])), (ra, rb)
    for i in range(3):
        tol = 2e-4 if i == 0 else 2e-2
        for k in snaps_a[i]:
            a, b = snaps_a[i][k], snaps_b[i][k]
            # (+ 2e-6: the batch-norm biases start at zero and are 1e-4 after two steps; the two float32 evaluations of
            # their gradients differ by up to 1e-6 of the largest, whichever vendor kernel PyTorch picks on the box)
            assert (a - b).abs().max().item() <= tol * max(b.abs().max().item(), 1e-3) + (2e-6 if i == 0 else 1e-4), (i, k, (a - b).abs().max().item())


def test_window_sampling_kernel():
    """az_sample_window (ReplayBufferSampler, alphazero/replay_buffer.py:175-195, for a window on the device): the
    batch is a function of (seed, draw counter), the launch increments the counter, the states come from the
    last min(count, capacity) only, are distinct while a batch fits into the window and repeat otherwise, every
    gathered row equals the window's row, and the draws are uniform over the window."""
    from alphazero_amd.games import ConnectFour
    from alphazero_amd.replay_buffer import DeviceReplayWindow
    dev = torch.device('cuda:0')
    win = DeviceReplayWindow(ConnectFour, 1000, dev)
    # fill the ring by hand: state i carries its own number in z and in one cell of x
    total = 1700                                               # the ring has wrapped: the last 1000 states are 700..1699
    for i0 in range(0, total, 850):
        idx = torch.arange(i0, i0 + 850, device=dev)
        slot = idx % win.capacity
        win.z[slot, 0] = idx.float()
        win.x[slot] = 0
        win.x[slot, 0, 0, 0] = idx.float()
        win.p[slot] = (idx.float() % 7).unsqueeze(1) + torch.arange(7, device=dev).float().unsqueeze(0)
        win.p_valid[slot] = ((idx.unsqueeze(1) + torch.arange(7, device=dev).unsqueeze(0)) % 3 == 0)
        win.count += 850
        win.count_dev.fill_(win.count)
    draw = torch.zeros(1, dtype=torch.int64, device=dev)
    b1 = win.sample(128, seed=5, draw_dev=draw)
    assert int(draw) == 1
    ids = b1['z'].reshape(-1).long()
    assert ids.min().item() >= 700 and ids.max().item() < 1700 and len(set(ids.tolist())) == 128
    assert torch.equal(b1['x'][:, 0, 0, 0].long(), ids) and float(b1['x'].sum()) == float(ids.sum())
    assert torch.equal(b1['p'], (ids.float() % 7).unsqueeze(1) + torch.arange(7, device=dev).float().unsqueeze(0))
    assert torch.equal(b1['p_valid'], (ids.unsqueeze(1) + torch.arange(7, device=dev).unsqueeze(0)) % 3 == 0)
    b2 = win.sample(128, seed=5, draw_dev=draw)                 # next draw: another batch
    assert int(draw) == 2 and not torch.equal(b2['z'], b1['z'])
    draw.zero_()
    again = win.sample(128, seed=5, draw_dev=draw)              # same (seed, draw): the same batch
    assert torch.equal(again['z'], b1['z'])
    other = win.sample(128, seed=6, draw_dev=torch.zeros(1, dtype=torch.int64, device=dev))
    assert not torch.equal(other['z'], b1['z'])
    big = win.sample(1500, seed=5, draw_dev=draw)               # larger than the window: with replacement
    assert len(set(big['z'].reshape(-1).long().tolist())) < 1500 and big['z'].min().item() >= 700
    # uniformity: 400 batches of 250 distinct states out of 1000 -> every state expected 100 times (sd 8.7)
    hits = torch.zeros(1700, device=dev)
    for _ in range(400):
        hits.index_add_(0, win.sample(250, seed=9, draw_dev=draw)['z'].reshape(-1).long(), torch.ones(250, device=dev))
    assert hits[:700].sum().item() == 0
    h = hits[700:]
    assert h.min().item() > 55 and h.max().item() < 145 and abs(h.std().item() - 8.7) < 1.5, (h.min(), h.max(), h.std())



def test_hand_written_training_steps_match_the_reference_scalars():
    """training_c4.json: three process_batch() calls of the REFERENCE's TrainingWorker on the CPU with
    AlphaZeroC4(2, 32) on four recorded Connect Four games.  The same seeds, games and configuration here with the
    hand-written pass (csrc/az_train_c4.hip), the HIP loss kernel and FlatSGD on the GPU: every logged scalar within
    1e-5 relative, the final weights and running statistics within 1e-4, the batch-norm counters equal."""
    from test_host_cpu import _golden_game
    from alphazero_amd.config import AlphaZeroConfig
    from alphazero_amd.replay_buffer import ReplayBufferDataset
    from alphazero_amd.training import TrainingWorker
    t = load_json('training_c4.json')
    cfg = AlphaZeroConfig(conf=t['conf'])
    ds = ReplayBufferDataset(cfg)
    for g in t['games']:
        ds.add_game(_golden_game(cfg, g), 1)

    class Server:
        def __init__(self):
            self.versions = []

        def update(self, model, version):
            self.versions.append(int(version))

    class Writer:
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, value, step):
            self.rows.append([tag, float(value), int(step)])

    torch.manual_seed(t['model_seed'])
    np.random.seed(t['sampler_seed'])
    server, writer = Server(), Writer()
    tw = TrainingWorker(cfg, server, ds, writer, 'cuda', 0)
    assert tw._hand is not None
    assert [tw.process_batch() for _ in range(3)] == t['returns'] and server.versions == t['versions']
    assert [(r[0], r[2]) for r in writer.rows] == [(r[0], r[2]) for r in t['scalars']]
    worst = 0.0
    for got, want in zip(writer.rows, t['scalars']):
        err = abs(got[1] - want[1]) / max(1.0, abs(want[1]))
        worst = max(worst, err)
        assert err <= 1e-5, (got, want)
    state = tw.model.state_dict()
    for k, h in t['final'].items():
        a, b = state[k].detach().cpu().numpy().reshape(-1), unhex(h, '<f4')
        assert np.allclose(a, b, rtol=1e-4, atol=1e-5), (k, np.abs(a - b).max())
    for k, v in t['counters'].items():
        assert int(state[k]) == v, k
    print('worst relative scalar error vs the reference:', worst)

