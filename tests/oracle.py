"""ctypes binding of oracle/libaz_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's algorithm
(oracle/az_oracle.c).  It is loaded here, by tests, by __graft_entry__.smoke()
and by bench.py's cpu_baseline leg, and nowhere in the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, 'oracle')
LIB_PATH = os.path.join(ORACLE_DIR, 'libaz_oracle.so')

TTT, C4 = 0, 1
GAME_ID = {'ttt': TTT, 'c4': C4}
MAX_ACTIONS, MAX_MOVES = 9, 42
WINNER_NONE = -2


class PuctEntry(C.Structure):
    _fields_ = [('cs', C.c_double), ('sqrt_ns', C.c_double)]


class Config(C.Structure):
    _fields_ = [('game', C.c_int), ('num_simulations', C.c_int), ('exploration_depth', C.c_int),
                ('reuse_search_tree', C.c_int), ('alpha', C.c_double), ('epsilon', C.c_double),
                ('c_base', C.c_double), ('c_init', C.c_double),
                ('lut', C.POINTER(PuctEntry)), ('lut_len', C.c_int)]


class Stats(C.Structure):
    _fields_ = [('sims', C.c_long), ('select_steps', C.c_long), ('leaf_evals', C.c_long),
                ('root_evals', C.c_long)]


class GameRecord(C.Structure):
    _fields_ = [('num_moves', C.c_int), ('winner', C.c_int),
                ('history', C.c_uint8 * MAX_MOVES), ('num_edges', C.c_uint8 * MAX_MOVES),
                ('visits', (C.c_uint32 * MAX_ACTIONS) * MAX_MOVES),
                ('root_W', (C.c_float * MAX_ACTIONS) * MAX_MOVES),
                ('root_P', (C.c_float * MAX_ACTIONS) * MAX_MOVES),
                ('action_index', C.c_uint8 * MAX_MOVES)]


class Rng(C.Structure):
    _fields_ = [('s', C.c_uint64 * 4)]


EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.c_int,
                      C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))
DIRICHLET_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_double))
UNIFORM_FN = C.CFUNCTYPE(C.c_double, C.c_void_p)

_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', ORACLE_DIR])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    u8p, ip, fp, dp = C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)
    L.orc_valid_actions.argtypes = [C.c_int, u8p, C.c_int, ip]
    L.orc_terminal.argtypes = [C.c_int, u8p, C.c_int]
    L.orc_winner.argtypes = [C.c_int, u8p, C.c_int]
    L.orc_terminal_value.argtypes = [C.c_int, u8p, C.c_int, C.c_int]
    L.orc_render.argtypes = [C.c_int, u8p, C.c_int, fp]
    L.orc_winning_positions.argtypes = [C.c_int, u8p]
    L.orc_position_hash.argtypes = [C.c_int, u8p, C.c_int]
    L.orc_position_hash.restype = C.c_uint32
    L.orc_node_new.argtypes = [C.c_void_p]
    L.orc_node_new.restype = C.c_void_p
    L.orc_node_free.argtypes = [C.c_void_p]
    L.orc_node_expanded.argtypes = [C.c_void_p]
    L.orc_node_num_edges.argtypes = [C.c_void_p]
    L.orc_node_expand.argtypes = [C.c_void_p, fp, C.c_int]
    L.orc_node_traverse.argtypes = [C.c_void_p, C.c_int]
    L.orc_node_traverse.restype = C.c_void_p
    L.orc_node_child.argtypes = [C.c_void_p, C.c_int]
    L.orc_node_child.restype = C.c_void_p
    L.orc_node_parent.argtypes = [C.c_void_p]
    L.orc_node_parent.restype = C.c_void_p
    L.orc_node_kill_siblings.argtypes = [C.c_void_p]
    L.orc_node_edges.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), fp, fp]
    L.orc_node_set_edges.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), fp, fp]
    L.orc_node_mean_value.argtypes = [C.c_void_p, fp]
    L.orc_node_ucb_score.argtypes = [C.c_void_p, C.c_double, C.c_double, C.POINTER(PuctEntry), C.c_int, dp]
    L.orc_node_pi.argtypes = [C.c_void_p, dp]
    L.orc_node_greedy_action.argtypes = [C.c_void_p]
    L.orc_node_sample_action.argtypes = [C.c_void_p, C.c_double]
    L.orc_node_add_noise.argtypes = [C.c_void_p, dp, C.c_double]
    L.orc_node_add_value_observation.argtypes = [C.c_void_p, C.c_int, C.c_float]
    L.orc_node_count.argtypes = [C.c_void_p]
    L.orc_node_count.restype = C.c_long
    L.orc_run_mcts.argtypes = [C.POINTER(Config), u8p, C.c_int, C.c_void_p, EVAL_FN, C.c_void_p,
                               DIRICHLET_FN, C.c_void_p, C.POINTER(Stats)]
    L.orc_play_game.argtypes = [C.POINTER(Config), EVAL_FN, C.c_void_p, DIRICHLET_FN, C.c_void_p,
                                UNIFORM_FN, C.c_void_p, C.POINTER(GameRecord), C.POINTER(Stats)]
    L.orc_rng_seed.argtypes = [C.POINTER(Rng), C.c_uint64]
    L.orc_rng_uniform.restype = C.c_double
    _lib = L
    return L


def _hist(history):
    arr = (C.c_uint8 * (MAX_MOVES + 1))(*[int(a) for a in history])
    return arr, len(history)


# ---- game rules --------------------------------------------------------
def num_actions(game):
    return lib().orc_num_actions(game)


def valid_actions(game, history):
    h, n = _hist(history)
    out = (C.c_int * MAX_ACTIONS)()
    k = lib().orc_valid_actions(game, h, n, out)
    return [out[i] for i in range(k)]


def terminal(game, history):
    h, n = _hist(history)
    return bool(lib().orc_terminal(game, h, n))


def winner(game, history):
    h, n = _hist(history)
    return lib().orc_winner(game, h, n)


def render(game, history):
    h, n = _hist(history)
    hh, ww = C.c_int(), C.c_int()
    lib().orc_board_shape(game, C.byref(hh), C.byref(ww))
    out = np.zeros((2, hh.value, ww.value), np.float32)
    lib().orc_render(game, h, n, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def winning_positions(game):
    buf = (C.c_uint8 * (69 * 42))()
    rows = lib().orc_winning_positions(game, buf)
    cols = 42 if game == C4 else 9
    return np.frombuffer(buf, np.uint8)[:rows * cols].reshape(rows, cols).copy()


def position_hash(game, history):
    h, n = _hist(history)
    return int(lib().orc_position_hash(game, h, n))


# ---- PUCT table ----------------------------------------------------------
def make_lut_array(lut_np):
    """lut_np: float64 [n,2] (Cs, sqrt(Ns)) -> ctypes array of PuctEntry."""
    n = lut_np.shape[0]
    arr = (PuctEntry * n)()
    C.memmove(arr, np.ascontiguousarray(lut_np, np.float64).ctypes.data, n * 16)
    return arr


# ---- tree wrapper ----------------------------------------------------------
class Node:
    """Thin handle on an orc_node; owner=True frees the subtree on delete."""

    def __init__(self, ptr=None, owner=None):
        L = lib()
        if ptr is None:
            ptr = L.orc_node_new(None)
            owner = True
        self.ptr = ptr
        self.owner = bool(owner)

    def __del__(self):
        if self.owner and self.ptr and _lib is not None:
            _lib.orc_node_free(self.ptr)
            self.ptr = None

    @property
    def expanded(self):
        return bool(lib().orc_node_expanded(self.ptr))

    @property
    def num_edges(self):
        return lib().orc_node_num_edges(self.ptr)

    def expand(self, p):
        p = np.ascontiguousarray(p, np.float32)
        lib().orc_node_expand(self.ptr, p.ctypes.data_as(C.POINTER(C.c_float)), len(p))

    def child(self, i):
        c = lib().orc_node_child(self.ptr, i)
        return Node(c, False) if c else None

    def traverse(self, i):
        return Node(lib().orc_node_traverse(self.ptr, i), False)

    def kill_siblings(self):
        lib().orc_node_kill_siblings(self.ptr)

    def edges(self):
        n = self.num_edges
        N, W, P = np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        if n:
            lib().orc_node_edges(self.ptr, N.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 W.ctypes.data_as(C.POINTER(C.c_float)),
                                 P.ctypes.data_as(C.POINTER(C.c_float)))
        return N, W, P

    def set_edges(self, N=None, W=None, P=None):
        def ptr(a, t, ct):
            return None if a is None else np.ascontiguousarray(a, t).ctypes.data_as(C.POINTER(ct))
        keep = [None if a is None else np.ascontiguousarray(a, t)
                for a, t in ((N, np.uint32), (W, np.float32), (P, np.float32))]
        lib().orc_node_set_edges(
            self.ptr,
            None if keep[0] is None else keep[0].ctypes.data_as(C.POINTER(C.c_uint32)),
            None if keep[1] is None else keep[1].ctypes.data_as(C.POINTER(C.c_float)),
            None if keep[2] is None else keep[2].ctypes.data_as(C.POINTER(C.c_float)))

    def mean_value(self):
        q = np.zeros(self.num_edges, np.float32)
        lib().orc_node_mean_value(self.ptr, q.ctypes.data_as(C.POINTER(C.c_float)))
        return q

    def ucb_score(self, lut=None, c_base=19652.0, c_init=1.25):
        s = np.zeros(self.num_edges, np.float64)
        lib().orc_node_ucb_score(self.ptr, c_base, c_init, lut, 0 if lut is None else len(lut),
                                 s.ctypes.data_as(C.POINTER(C.c_double)))
        return s

    def pi(self):
        s = np.zeros(self.num_edges, np.float64)
        lib().orc_node_pi(self.ptr, s.ctypes.data_as(C.POINTER(C.c_double)))
        return s

    def greedy_action(self):
        return lib().orc_node_greedy_action(self.ptr)

    def sample_action(self, u):
        return lib().orc_node_sample_action(self.ptr, float(u))

    def add_noise(self, eta, eps):
        eta = np.ascontiguousarray(eta, np.float64)
        lib().orc_node_add_noise(self.ptr, eta.ctypes.data_as(C.POINTER(C.c_double)), float(eps))

    def count(self):
        return lib().orc_node_count(self.ptr)

    def dump(self, path=()):
        """Canonical listing [(path, N, W, P)] of every expanded node, DFS by slot."""
        out = []
        if not self.expanded:
            return out
        N, W, P = self.edges()
        out.append((tuple(path), N, W, P))
        for i in range(self.num_edges):
            c = self.child(i)
            if c is not None:
                out.extend(c.dump(tuple(path) + (i,)))
        return out


# ---- evaluators / randomness providers ----------------------------------------
def eval_uniform():
    return C.cast(lib().orc_eval_uniform, EVAL_FN)


def eval_subset():
    return C.cast(lib().orc_eval_subset, EVAL_FN)


def eval_from_table(game, table):
    """table: dict render-bitstring -> (p float32[nvalid], v float32).  Raises on a miss."""
    misses = []

    def fn(ctx, g, hist, n, valid, nvalid, p_out, v_out):
        x = render(g, [hist[i] for i in range(n)])
        key = ''.join('1' if t else '0' for t in x.reshape(-1))
        if key not in table:
            misses.append(key)
            for i in range(nvalid):
                p_out[i] = 1.0 / nvalid
            v_out[0] = 0.0
            return
        p, v = table[key]
        for i in range(nvalid):
            p_out[i] = p[i]
        v_out[0] = v
    cb = EVAL_FN(fn)
    cb.misses = misses
    return cb


def eval_from_python(fn_py):
    """fn_py(game, history, valid) -> (p, v)."""
    def fn(ctx, g, hist, n, valid, nvalid, p_out, v_out):
        p, v = fn_py(g, [hist[i] for i in range(n)], [valid[i] for i in range(nvalid)])
        for i in range(nvalid):
            p_out[i] = p[i]
        v_out[0] = v
    return EVAL_FN(fn)


def dirichlet_from_list(etas):
    it = iter(etas)

    def fn(ctx, alpha, n, out):
        eta = next(it)
        assert len(eta) == n, (len(eta), n)
        for i in range(n):
            out[i] = eta[i]
    return DIRICHLET_FN(fn)


def uniform_from_list(us):
    it = iter(us)
    return UNIFORM_FN(lambda ctx: float(next(it)))


def make_config(game, S, depth, reuse, lut, alpha=0.3, epsilon=0.25, c_base=19652.0, c_init=1.25):
    cfg = Config(game, S, depth, int(reuse), alpha, epsilon, c_base, c_init,
                 C.cast(lut, C.POINTER(PuctEntry)) if lut is not None else None,
                 0 if lut is None else len(lut))
    cfg._keep = lut
    return cfg


def run_mcts(cfg, history, root, eval_cb, dir_cb=None, stats=None):
    h, n = _hist(history)
    lib().orc_run_mcts(C.byref(cfg), h, n, root.ptr, eval_cb, None,
                       dir_cb if dir_cb is not None else C.cast(None, DIRICHLET_FN), None,
                       C.byref(stats) if stats is not None else None)


def play_game(cfg, eval_cb, dir_cb, uni_cb, stats=None):
    rec = GameRecord()
    lib().orc_play_game(C.byref(cfg), eval_cb, None, dir_cb, None, uni_cb, None, C.byref(rec),
                        C.byref(stats) if stats is not None else None)
    return rec


def play_game_builtin_rng(cfg, eval_cb, seed, stats=None):
    """Timing/statistics runs: the oracle's own xoshiro + Marsaglia-Tsang provider."""
    L = lib()
    rng = Rng()
    L.orc_rng_seed(C.byref(rng), seed)
    rec = GameRecord()
    L.orc_play_game(C.byref(cfg), eval_cb, None,
                    C.cast(L.orc_rng_dirichlet, DIRICHLET_FN), C.byref(rng),
                    C.cast(L.orc_rng_uniform, UNIFORM_FN), C.byref(rng), C.byref(rec),
                    C.byref(stats) if stats is not None else None)
    return rec
