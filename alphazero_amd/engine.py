"""SelfPlayEngine: Python handle on one az_engine (include/azhip.h).

Owns the I/O tensors (torch, on the engine's device) and exposes each C-ABI entry
point as a method.  All work is stream-ordered on torch's current stream.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import (AzError, EngineConfig, GameRecord, Stats, LEAF_EVAL, LEAF_NONE, LEAF_ROOT,
                   LEAF_TERMINAL, NO_NODE, check, ptr)

GAME_SHAPES = {_lib.GAME_TTT: (9, 3, 3, 9), _lib.GAME_C4: (7, 6, 7, 42),      # A, H, W, max moves
               _lib.GAME_GENERIC: (15, 0, 0, 0)}    # user-defined games: up to 15 actions, rules on the host


def make_puct_lut(n, c_base=19652, c_init=1.25):
    """float64 [n,2] table {Cs, sqrt(Ns)} for Ns = 0..n-1.

    The two expressions are the reference's own (alphazero/mcts.py:55-57), evaluated
    by the host's numpy on uint64 scalars, so the engine uses exactly the log the
    reference would have computed on this machine."""
    ns = np.arange(n, dtype=np.uint64)
    cs = np.log((1 + ns + c_base) / c_base) + c_init
    return np.ascontiguousarray(np.stack([cs, np.sqrt(ns)], axis=1), dtype=np.float64)


def _stream(device=None):
    """torch's current stream on `device` (default: the current device) as the void* the C ABI takes."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class SelfPlayEngine:
    def __init__(self, game, num_games, num_simulations, exploration_depth=0, reuse_search_tree=True,
                 alpha=0.3, epsilon=0.25, c_base=19652, c_init=1.25, seed=0, device=0,
                 node_capacity=0, model_version=0, first_game_uid=0, uid_stride=0, lut=None):
        if not torch.cuda.is_available():
            raise AzError('SelfPlayEngine needs a GPU: there is no CPU path in this package')
        self.lib = _lib.load()
        self.game = int(game)
        self.A, self.H, self.W, self.max_moves = GAME_SHAPES[self.game]
        self.G = int(num_games)
        self.S = int(num_simulations)
        self.device = torch.device('cuda', device)
        self.cfg = EngineConfig(self.game, self.G, int(node_capacity), self.S, int(exploration_depth),
                                int(bool(reuse_search_tree)), int(device), int(model_version),
                                float(alpha), float(epsilon), float(c_base), float(c_init),
                                int(seed), int(first_game_uid), int(uid_stride))
        h = C.c_void_p()
        check(self.lib.az_engine_create(C.byref(self.cfg), C.byref(h)))
        self.h = h
        if lut is None:
            reach = self.S * (self.max_moves if reuse_search_tree else 1) + 2
            lut = make_puct_lut(max(reach, 1024), c_base, c_init)
        self.set_puct_lut(lut)
        with torch.cuda.device(self.device):
            self.leaf_bb = torch.zeros(self.G, 2, dtype=torch.int64, device=self.device)
            self.leaf_info = torch.zeros(self.G, dtype=torch.int32, device=self.device)
            self.out_action = torch.zeros(self.G, dtype=torch.int32, device=self.device)
            self.out_nedges = torch.zeros(self.G, dtype=torch.int32, device=self.device)
            self.out_visits = torch.zeros(self.G, self.A, dtype=torch.int32, device=self.device)
            self.out_winner = torch.zeros(self.G, dtype=torch.int32, device=self.device)
        self._drain_buf = None
        self._take_buf = None
        self._pv = None

    def _stream(self):
        return _stream(self.device)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.az_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- configuration ---------------------------------------------------------
    def set_puct_lut(self, lut):
        lut = np.ascontiguousarray(lut, dtype=np.float64)
        assert lut.ndim == 2 and lut.shape[1] == 2
        check(self.lib.az_set_puct_lut(self.h, C.c_void_p(lut.ctypes.data), lut.shape[0]))
        self.lut_len = int(lut.shape[0])

    def ensure_puct_lut(self, reach):
        """Make sure the PUCT table has a row for every visit sum up to `reach` (the device computes no
        logarithm: past the table's end it would reuse the last row and raise error bit 3)."""
        if reach > self.lut_len:
            self.set_puct_lut(make_puct_lut(max(int(reach), 2 * self.lut_len), self.cfg.c_base, self.cfg.c_init))

    def set_model_version(self, v):
        check(self.lib.az_set_model_version(self.h, int(v)))

    def reset_games(self):
        check(self.lib.az_reset_games(self.h, self._stream()))

    def abandon_games(self):
        """Give up every game in progress; the slots start their next games (new uids) under the current model version."""
        check(self.lib.az_abandon_games(self.h, self._stream()))

    def enable_eval_cache(self, log2_entries):
        """Device table of 2**log2_entries evaluator outputs keyed by (position, model version); 0
        turns it off.  sim_step then answers repeated leaves itself (LEAF_CACHED) and writes their rows
        of p and v.  Only for evaluators whose rows are pure functions of the position (the one-launch
        HIP networks are); the reference has no such cache (alphazero/mcts.py:139-142)."""
        check(self.lib.az_eval_cache_enable(self.h, int(log2_entries)))

    def eval_cache_hits(self):
        n = C.c_uint64()
        check(self.lib.az_eval_cache_hits(self.h, C.byref(n), self._stream()))
        return int(n.value)

    # ---- one move, lock-step ---------------------------------------------------
    def begin_move(self):
        check(self.lib.az_begin_move(self.h, ptr(self.leaf_bb), ptr(self.leaf_info), self._stream()))
        return self.leaf_bb, self.leaf_info

    def apply_root(self, p, eta=None):
        assert p.dtype == torch.float32 and p.is_contiguous() and p.shape == (self.G, self.A)
        if eta is not None:
            assert eta.dtype == torch.float64 and eta.is_contiguous() and eta.shape == (self.G, self.A)
        check(self.lib.az_apply_root(self.h, ptr(p), ptr(eta), self._stream()))

    def begin_move_uniform(self, eta=None):
        if eta is not None:
            assert eta.dtype == torch.float64 and eta.is_contiguous() and eta.shape == (self.G, self.A)
        check(self.lib.az_begin_move_uniform(self.h, ptr(eta), self._stream()))

    def select(self):
        check(self.lib.az_select(self.h, ptr(self.leaf_bb), ptr(self.leaf_info), self._stream()))
        return self.leaf_bb, self.leaf_info

    def _check_pv(self, p, v):
        assert p.dtype == torch.float32 and p.is_contiguous() and p.shape == (self.G, self.A)
        assert v.dtype == torch.float32 and v.is_contiguous() and v.numel() == self.G

    def expand_backup(self, p, v):
        self._check_pv(p, v)
        check(self.lib.az_expand_backup(self.h, ptr(p), ptr(v), self._stream()))

    def sim_step(self, p=None, v=None):
        if p is not None:
            self._check_pv(p, v)
        check(self.lib.az_sim_step(self.h, ptr(p), ptr(v), ptr(self.leaf_bb), ptr(self.leaf_info), self._stream()))
        return self.leaf_bb, self.leaf_info

    def fused_max_games(self):
        """Largest engine that move_fused accepts on this device (64 games per compute unit)."""
        return int(self.lib.az_c4_move_fused_max_games(self.h))

    def limit_fused_workgroups(self, max_workgroups):
        """Cap the workgroups move_fused spreads the games over (0 = one per compute unit, the default)."""
        check(self.lib.az_c4_move_fused_limit_workgroups(self.h, int(max_workgroups)))

    def fused_counters(self):
        """What move_fused's workgroups did since the last call: network tasks with two boards / one board,
        head batches, boards in head batches (synchronises; resets the counters)."""
        out = (C.c_uint64 * 4)()
        check(self.lib.az_c4_move_fused_counters(self.h, out, self._stream()))
        return {'pair_tasks': int(out[0]), 'single_tasks': int(out[1]), 'head_batches': int(out[2]),
                'head_boards': int(out[3])}

    def move_fused(self, net):
        """All S simulations of the move in one launch with a FusedC4Net or a FusedC4NetHX (after begin_move /
        apply_root)."""
        if self._pv is None:
            self._pv = (torch.zeros(self.G, self.A, dtype=torch.float32, device=self.device),
                        torch.zeros(self.G, dtype=torch.float32, device=self.device))
        p, v = self._pv
        # a FusedC4NetHX (fp16-pair operands, fp32-grade outputs) goes to the kernel's form for that blob
        fn = self.lib.az_c4_move_fused_hx if getattr(net, 'precision', 'bf16') == 'hx' else self.lib.az_c4_move_fused
        check(fn(self.h, ptr(net.blob), int(net.num_blocks), ptr(self.leaf_bb), ptr(self.leaf_info), ptr(p), ptr(v),
                 self._stream()))

    def move_fused_static(self, net):
        """move_fused through the entry point that allocates nothing and takes no staging buffers
        (az_c4_move_fused_static: up to 16384 games, one shared staging area per GPU)."""
        check(self.lib.az_c4_move_fused_static(self.h, ptr(net.blob), int(net.num_blocks), self._stream()))

    def run_sims_uniform(self, num_sims):
        check(self.lib.az_run_sims_uniform(self.h, int(num_sims), self._stream()))

    def choose_and_advance(self, u=None):
        if u is not None:
            assert u.dtype == torch.float64 and u.is_contiguous() and u.numel() == self.G
        check(self.lib.az_choose_and_advance(self.h, ptr(u), ptr(self.out_action), ptr(self.out_nedges),
                                             ptr(self.out_visits), ptr(self.out_winner), self._stream()))
        return self.out_action, self.out_nedges, self.out_visits, self.out_winner

    # ---- results -----------------------------------------------------------------
    def drain_finished_bytes(self):
        """Finished games since the last drain as a uint8 array [n, sizeof(az_game_record)] (a view of
        a pinned host buffer that the next drain overwrites; synchronises).  The raw form is what
        travels between ranks (distributed.gather_record_bytes)."""
        cap = 2 * self.G + 16
        if self._drain_buf is None:
            self._drain_buf = torch.empty(cap * C.sizeof(GameRecord), dtype=torch.uint8).pin_memory()
        n = C.c_int32()
        check(self.lib.az_drain_finished(self.h, C.cast(self._drain_buf.data_ptr(), C.POINTER(GameRecord)), cap,
                                         C.byref(n), self._stream()))
        return self._drain_buf.numpy()[:n.value * C.sizeof(GameRecord)].reshape(n.value, C.sizeof(GameRecord))

    def take_finished_device(self):
        """Finished games since the last hand-over, left on the device: (records uint8 [ring capacity,
        sizeof(az_game_record)], head int32[2] = {number of valid records, error bits}), both tensors of
        this engine that the next call overwrites.  No host synchronisation."""
        cap = 2 * self.G + 16
        if self._take_buf is None:
            self._take_buf = (torch.zeros(cap, C.sizeof(GameRecord), dtype=torch.uint8, device=self.device),
                              torch.zeros(2, dtype=torch.int32, device=self.device))
        recs, head = self._take_buf
        check(self.lib.az_take_finished_device(self.h, ptr(recs), cap, ptr(head), self._stream()))
        return recs, head

    def drain_finished(self):
        """Finished games since the last drain as a list of GameRecord (synchronises)."""
        raw = self.drain_finished_bytes()
        return [GameRecord.from_buffer_copy(raw[i].tobytes()) for i in range(raw.shape[0])]

    def error_flags(self):
        """Error bits seen by the drains (and get_tree) since the last reset: 1 = a node pool ran out, 2 =
        finished games were dropped because the record ring was full, 8 = a visit sum beyond the PUCT table,
        16 = move_fused abandoned a launch, 32 = a user-defined game's search path deeper than 48 levels.
        No device work."""
        return int(self.lib.az_engine_error_flags(self.h))

    def stats(self):
        st = Stats()
        check(self.lib.az_get_stats(self.h, C.byref(st), self._stream()))
        return {k: int(getattr(st, k)) for k, _ in Stats._fields_}

    # ---- user-defined games: tree on the device, rules with the caller ----------------
    def generic_root(self, g, p=None, eta=None):
        """Expand slot g's root with priors p (when it has no node yet) and blend the noise eta (float64)."""
        p = None if p is None else np.ascontiguousarray(p, np.float32)
        if eta is not None:
            full = np.zeros(15, np.float64)
            full[:len(eta)] = eta
            eta = full
        check(self.lib.az_generic_root(self.h, int(g), 0 if p is None else len(p),
                                       None if p is None else C.c_void_p(p.ctypes.data),
                                       None if eta is None else C.c_void_p(eta.ctypes.data), self._stream()))

    def generic_select(self, g):
        """Action indices from the root to the leaf of slot g's next simulation (synchronises)."""
        depth = C.c_int32()
        slots = (C.c_uint8 * 48)()
        check(self.lib.az_generic_select(self.h, int(g), C.byref(depth), slots, self._stream()))
        return [int(slots[i]) for i in range(depth.value)]

    def generic_expand_backup(self, g, p, v_edge):
        """p: priors of the leaf's legal moves (None / empty for a terminal leaf); v_edge: value for the leaf's edge."""
        p = None if p is None or len(p) == 0 else np.ascontiguousarray(p, np.float32)
        check(self.lib.az_generic_expand_backup(self.h, int(g), 0 if p is None else len(p),
                                                None if p is None else C.c_void_p(p.ctypes.data),
                                                C.c_float(float(np.asarray(v_edge, np.float32).reshape(-1)[0])), self._stream()))

    # ---- tree transfer (single-game mirror) -------------------------------------------
    @property
    def node_words(self):
        return 4 * (self.A + 1)

    def set_position(self, g, bb_next, bb_prev, num_moves, nodes=None, root_node=NO_NODE):
        torch.cuda.synchronize(self.device)
        if nodes is None or len(nodes) == 0:
            check(self.lib.az_set_position(self.h, g, bb_next, bb_prev, num_moves, None, 0, NO_NODE))
            return
        nodes = np.ascontiguousarray(nodes, dtype=np.uint32).reshape(-1, self.node_words)
        check(self.lib.az_set_position(self.h, g, bb_next, bb_prev, num_moves,
                                       C.c_void_p(nodes.ctypes.data), nodes.shape[0], int(root_node)))

    def get_tree(self, g):
        """(uint32 [n, A+1, 4] node image, root index) of game slot g."""
        torch.cuda.synchronize(self.device)
        n, root = C.c_int32(), C.c_uint32()
        check(self.lib.az_get_tree(self.h, g, None, 0, C.byref(n), C.byref(root)))
        nodes = np.zeros((max(n.value, 1), self.node_words), np.uint32)
        check(self.lib.az_get_tree(self.h, g, C.c_void_p(nodes.ctypes.data), nodes.shape[0],
                                   C.byref(n), C.byref(root)))
        return nodes[:n.value].reshape(n.value, self.A + 1, 4), int(root.value)


# ---- stateless bulk game kernels ---------------------------------------------------
def positions_from_history(game, histories, device='cuda'):
    """List of action lists -> (bb int64[n,2], num_moves int32[n]) on the device."""
    lib = _lib.load()
    n = len(histories)
    stride = max([len(h) for h in histories] + [1])
    hist = np.zeros((n, stride), np.uint8)
    lens = np.zeros(n, np.int32)
    for i, h in enumerate(histories):
        hist[i, :len(h)] = h
        lens[i] = len(h)
    dh = torch.from_numpy(hist).to(device)
    dl = torch.from_numpy(lens).to(device)
    bb = torch.zeros(n, 2, dtype=torch.int64, device=device)
    nm = torch.zeros(n, dtype=torch.int32, device=device)
    check(lib.az_positions_from_history(game, ptr(dh), ptr(dl), stride, n, ptr(bb), ptr(nm), _stream()))
    return bb, nm


def game_query(game, bb, nm):
    lib = _lib.load()
    n = bb.shape[0]
    legal = torch.zeros(n, dtype=torch.int32, device=bb.device)
    term = torch.zeros_like(legal)
    win = torch.zeros_like(legal)
    check(lib.az_game_query(game, ptr(bb), ptr(nm), n, ptr(legal), ptr(term), ptr(win), _stream()))
    return legal, term, win


def game_step(game, bb, nm, action):
    lib = _lib.load()
    n = bb.shape[0]
    status = torch.zeros(n, dtype=torch.int32, device=bb.device)
    check(lib.az_game_step(game, ptr(bb), ptr(nm), ptr(action), n, ptr(status), _stream()))
    return status


def render(game, bb):
    """Network input planes x float32[n,2,H,W] and legal mask bool[n,A] for positions bb."""
    lib = _lib.load()
    A, H, W, _ = GAME_SHAPES[game]
    n = bb.shape[0]
    x = torch.empty(n, 2, H, W, dtype=torch.float32, device=bb.device)
    valid = torch.empty(n, A, dtype=torch.uint8, device=bb.device)
    check(lib.az_render(game, ptr(bb), n, ptr(x), ptr(valid), _stream()))
    return x, valid.bool()


def masked_softmax(game, logits, leaf_info):
    lib = _lib.load()
    logits = logits.contiguous().float()
    p = torch.empty_like(logits)
    check(lib.az_masked_softmax(game, ptr(logits), ptr(leaf_info), logits.shape[0], ptr(p), _stream()))
    return p


def leaf_kind(leaf_info):
    return leaf_info & 0xFF


def leaf_legal(leaf_info):
    return (leaf_info >> 16) & 0x1FF
