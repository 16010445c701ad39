"""MCTreeNode / evaluate / run_mcts with the reference's call signatures
(alphazero/mcts.py:5-158), executed by the HIP engine.

`run_mcts(game, root, model, num_simulations, alpha, epsilon)` mutates `root` in
place exactly like the reference: the Python tree under `root` is flattened into
the engine's node image, the simulations run on the GPU (select / expand /
backup kernels, one simulation per launch with the model called in between) and
the image is read back into the same Python objects.  The Dirichlet draw is
taken from numpy's global RandomState, as the reference does, so seeded runs
reproduce the reference's numbers bit for bit.

There is no CPU implementation of the search here: without the HIP library or
a GPU, `run_mcts` raises.
"""
import numpy as np
import torch

from . import _lib, engine as _engine
from ._lib import LEAF_EVAL, LEAF_ROOT, NO_NODE


class MCTreeNode:
    """Search-tree node; statistics of the edges to its children live here."""

    EDGE_DTYPE = [('N', np.uint32), ('W', np.float32), ('P', np.float32)]

    def __init__(self, parent=None):
        self.parent = parent
        self.children = None
        self._edges = None

    @property
    def expanded(self):
        return self.children is not None

    def expand(self, p):
        p = np.asarray(p)
        assert np.all(p >= 0), f'Node must be expanded with proper, non-negative distribution. Got: {p}'
        assert abs(float(np.sum(p)) - 1) < 1e-6, f'Node must be expanded with normalized distribution. Got: {p}'
        self.children = [None] * len(p)
        self._edges = np.zeros(len(p), dtype=MCTreeNode.EDGE_DTYPE)
        self._edges['P'] = p

    def traverse(self, action_index):
        assert action_index < len(self.children)
        child = self.children[action_index]
        if child is None:
            child = self.children[action_index] = MCTreeNode(self)
        return child

    @property
    def action_prior(self):
        return self._edges['P']

    @property
    def num_visits(self):
        return self._edges['N']

    @property
    def mean_value(self):
        """Q(s,a) in float32: W/N formed in double and rounded once, 0 where N == 0."""
        N, W = self._edges['N'], self._edges['W']
        q = np.zeros(len(N), np.float32)
        nz = N != 0
        q[nz] = (W[nz].astype(np.float64) / N[nz].astype(np.float64)).astype(np.float32)
        return q

    def ucb_score(self, c_base=19652, c_init=1.25):
        """PUCT score in float64 with the engine's operation order:
        (double)Q + (Cs * (double)P) * (sqrt(Ns) / (double)(1 + N))."""
        N = self._edges['N']
        ns = np.uint64(N.sum(dtype=np.uint64))
        cs = np.log((1 + ns + c_base) / c_base) + c_init
        s = np.sqrt(ns) / (N.astype(np.float64) + 1.0)
        u = (np.float64(cs) * self._edges['P'].astype(np.float64)) * s
        return self.mean_value.astype(np.float64) + u

    def pi(self, temperature=1):
        assert temperature > 0
        nt = self._edges['N'].astype(np.float64) ** (1 / temperature)
        total = nt.sum()
        if total > 0:
            return nt / total
        return np.full(len(nt), 1.0 / len(nt))

    def sample_action(self, temperature=1):
        return np.random.choice(len(self.children), p=self.pi(temperature))

    def greedy_action(self):
        return int(np.argmax(self.pi()))

    def kill_siblings(self):
        if self.parent is None:
            return
        sibs = self.parent.children
        for i, s in enumerate(sibs):
            if s is not self:
                sibs[i] = None

    def add_value_observation(self, action_index, v):
        self._edges['W'][action_index] += v
        self._edges['N'][action_index] += 1

    def add_exploration_noise(self, alpha, epsilon):
        eta = np.random.dirichlet([alpha] * len(self._edges))
        self._blend_noise(eta, epsilon)

    def _blend_noise(self, eta, epsilon):
        a = np.float32(1 - epsilon) * self._edges['P']                       # float32
        self._edges['P'] = (a.astype(np.float64) + epsilon * eta).astype(np.float32)


# --------------------------------------------------------------------------
# model evaluation on engine leaves
# --------------------------------------------------------------------------
def _model_device(model):
    try:
        return next(model.parameters()).device
    except Exception:
        return torch.device('cpu')


def evaluate_leaves(game_id, model, leaf_bb, leaf_info):
    """Run `model` on a batch of engine leaves; returns (p float32[n,A] action-indexed
    soft-maxed over the legal logits, v float32[n]) on the leaves' device.

    Mirrors alphazero/mcts.py:94-107 for a batch: planes and the legal mask are
    rendered on the GPU (az_render); the soft-max is az_masked_softmax."""
    model.eval()
    with torch.no_grad():
        x, valid = _engine.render(game_id, leaf_bb)
        mdev = _model_device(model)
        logits, v = model(x.to(mdev), valid.to(mdev))
        logits = logits.to(leaf_bb.device, torch.float32).reshape(x.shape[0], -1).contiguous()
        v = v.to(leaf_bb.device, torch.float32).reshape(-1).contiguous()
        p = _engine.masked_softmax(game_id, logits, leaf_info)
    return p, v


def evaluate(game, model):
    """alphazero/mcts.py:94-107: (p float32[num_legal], v float32[1]) for one game object.

    The soft-max over the legal logits is torch's, on the model's device, exactly
    as the reference calls it, so the numbers are the reference's for the same model."""
    model.eval()
    with torch.no_grad():
        dev = _model_device(model)
        x = torch.from_numpy(game.render()[None, ...]).to(dev)
        valid = torch.zeros(1, game.NUM_ACTIONS, dtype=torch.bool, device=dev)
        valid[0, game.valid_actions] = 1
        p, v = model(x, valid)
        p = torch.nn.functional.softmax(p[valid], dim=0)
        return p.cpu().numpy(), v[0].cpu().numpy()


# --------------------------------------------------------------------------
# tree <-> node image
# --------------------------------------------------------------------------
def _flatten(root, A):
    """BFS over expanded nodes below `root` -> (image uint32[n, A+1, 4], [node objects])."""
    if not root.expanded:
        return np.zeros((0, A + 1, 4), np.uint32), []
    order, index = [root], {id(root): 0}
    i = 0
    while i < len(order):
        for c in order[i].children:
            if c is not None and c.expanded:
                index[id(c)] = len(order)
                order.append(c)
        i += 1
    img = np.zeros((len(order), A + 1, 4), np.uint32)
    img[:, :A, 3] = NO_NODE
    for k, node in enumerate(order):
        n = len(node.children)
        e = node._edges
        img[k, :n, 0] = e['N']
        img[k, :n, 1] = e['W'].view(np.uint32)
        img[k, :n, 2] = e['P'].view(np.uint32)
        for j, c in enumerate(node.children):
            if c is not None and c.expanded:
                img[k, j, 3] = index[id(c)]
        img[k, A, 0] = n
        img[k, A, 1] = int(e['N'].sum(dtype=np.uint64))
    return img, order


def _unflatten(img, objs, root, root_idx, A):
    """Write the engine's image back into the Python tree (new nodes get new objects)."""
    n = img.shape[0]
    objs = list(objs) + [None] * (n - len(objs))
    if root_idx != NO_NODE and objs[root_idx] is None:
        objs[root_idx] = root
    # parents first (children always have larger indices than the node that links them,
    # except images uploaded by _flatten, which are BFS-ordered: also parent-first)
    for k in range(n):
        node = objs[k]
        if node is None:
            continue
        ne = int(img[k, A, 0])
        if not node.expanded:
            node.children = [None] * ne
            node._edges = np.zeros(ne, dtype=MCTreeNode.EDGE_DTYPE)
        node._edges['N'] = img[k, :ne, 0]
        node._edges['W'] = img[k, :ne, 1].view(np.float32)
        node._edges['P'] = img[k, :ne, 2].view(np.float32)
        for j in range(ne):
            c = int(img[k, j, 3])
            if c == NO_NODE:
                continue
            if objs[c] is None:
                child = node.children[j]
                if child is None:
                    child = node.children[j] = MCTreeNode(node)
                objs[c] = child


_ENGINES = {}


def _single_engine(game_id, alpha, epsilon, need_nodes):
    key = (game_id, torch.cuda.current_device())
    eng = _ENGINES.get(key)
    cap = max(1 << 14, 2 * need_nodes)
    if eng is None or eng.cfg.node_capacity < need_nodes or eng.cfg.alpha != alpha or eng.cfg.epsilon != epsilon:
        if eng is not None:
            eng.close()
        eng = _engine.SelfPlayEngine(game_id, 1, 0, alpha=alpha, epsilon=epsilon, node_capacity=cap,
                                     device=torch.cuda.current_device(),
                                     lut=_engine.make_puct_lut(1 << 16))
        _ENGINES[key] = eng
    return eng


def _run_mcts_generic(game, root, model, num_simulations, alpha, epsilon):
    """run_mcts for a Game subclass the device has no rules for (alphazero/test/test_mcts.py:107-124 is
    one): the tree, PUCT select, expansion and backup run on the engine (az_generic_*), every rule query
    (valid_actions, take_action, terminal, terminal_value, render) goes to the caller's Game object,
    exactly where alphazero/mcts.py:119-145 makes them."""
    A = 15
    if game.NUM_ACTIONS > A:
        raise TypeError(f'run_mcts: user-defined games may have up to {A} actions, got {game.NUM_ACTIONS}')
    img, objs = _flatten(root, A)
    eng = _single_engine(_lib.GAME_GENERIC, float(alpha), float(epsilon), img.shape[0] + num_simulations + 2)
    eng.ensure_puct_lut(_visits_below(root) + num_simulations + 2)
    eng.set_position(0, 0, 0, len(game), img if len(objs) else None, 0 if len(objs) else NO_NODE)
    # mcts.py:114-117
    p_root = None
    if not root.expanded:
        p_root, _ = evaluate(game, model)
        n_edges = len(p_root)
    else:
        n_edges = len(root.children)
    eng.generic_root(0, p_root, np.random.dirichlet([alpha] * n_edges))
    for _ in range(num_simulations):
        simulation = game.clone()
        try:
            path = eng.generic_select(0)                       # mcts.py:123-127
        except _lib.AzError:
            eng.reset_games()                                  # clears the engine's error word; `root` is untouched
            raise
        for ai in path:
            simulation.take_action(simulation.valid_actions[ai])
        if simulation.terminal:                                # mcts.py:131-134
            eng.generic_expand_backup(0, None, simulation.terminal_value(simulation.prev_player))
        else:                                                  # mcts.py:135-144
            p, v = evaluate(simulation, model)
            assert np.all(p >= 0), f'Node must be expanded with proper, non-negative distribution. Got: {p}'
            assert abs(float(np.sum(p)) - 1) < 1e-6, f'Node must be expanded with normalized distribution. Got: {p}'
            eng.generic_expand_backup(0, p, -np.float32(v))
    st_img, root_idx = eng.get_tree(0)
    _check_flags(eng)
    _unflatten(st_img, objs, root, root_idx, A)


def _visits_below(root):
    return int(np.sum(root.num_visits)) if root.expanded else 0


def _check_flags(eng):
    flags = eng.error_flags()
    if flags:
        eng.reset_games()            # (clears the flag word; the slot is loaded afresh by the next search)
        raise _lib.AzError(f'run_mcts: the engine reported error bits {flags:#x} (1 = node pool exhausted, '
                           '8 = visit sum beyond the PUCT table, 32 = search path deeper than 48 levels)')


def run_mcts(game, root, model, num_simulations, alpha=0.3, epsilon=0.25):
    """alphazero/mcts.py:110-158 on the GPU; mutates the tree under `root` in place."""
    if not torch.cuda.is_available():
        raise _lib.AzError('run_mcts needs a GPU: there is no CPU search path in this package')
    game_id = getattr(game, 'AZ_GAME_ID', None)
    if game_id is None:
        return _run_mcts_generic(game, root, model, num_simulations, alpha, epsilon)
    A = game.NUM_ACTIONS
    img, objs = _flatten(root, A)
    eng = _single_engine(game_id, float(alpha), float(epsilon), img.shape[0] + num_simulations + 2)
    nx, pv, nm = game.position()
    eng.ensure_puct_lut(_visits_below(root) + num_simulations + 2)
    eng.set_position(0, nx, pv, nm, img if len(objs) else None, 0 if len(objs) else NO_NODE)
    dev = eng.device

    # mcts.py:114-117: expand the root if needed, then noise from numpy's global RNG
    leaf_bb, leaf_info = eng.begin_move()
    kind = int(leaf_info[0].item()) & 0xFF
    p = torch.zeros(1, A, dtype=torch.float32, device=dev)
    if kind == LEAF_ROOT:
        pc, _ = evaluate(game, model)
        p[0, game.valid_actions] = torch.from_numpy(np.asarray(pc, np.float32)).to(dev)
        n_edges = len(pc)
    elif game.terminal:
        return
    else:
        n_edges = len(root.children)
    eta = torch.zeros(1, A, dtype=torch.float64, device=dev)
    eta[0, :n_edges] = torch.from_numpy(np.random.dirichlet([alpha] * n_edges)).to(dev)
    eng.apply_root(p, eta)

    v = torch.zeros(1, dtype=torch.float32, device=dev)
    for _ in range(num_simulations):
        leaf_bb, leaf_info = eng.select()
        info = int(leaf_info[0].item())
        if (info & 0xFF) != LEAF_EVAL:
            continue
        # mcts.py:137: one leaf through the model; soft-max over the legal logits by torch
        model.eval()
        with torch.no_grad():
            x, valid = _engine.render(game_id, leaf_bb)
            mdev = _model_device(model)
            logits, val = model(x.to(mdev), valid.to(mdev))
            vm = valid.to(logits.device)
            pc = torch.nn.functional.softmax(logits[vm], dim=0)
            p.zero_()
            p[0, valid[0]] = pc.to(dev, torch.float32)
            v[0] = val.reshape(-1)[0].to(dev, torch.float32)
        eng.expand_backup(p, v)
    st_img, root_idx = eng.get_tree(0)
    _check_flags(eng)
    _unflatten(st_img, objs, root, root_idx, A)
