"""ctypes binding of libazhip.so (the HIP engine).  No fallback: if the library is
missing or fails to load, importing the engine raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# AZHIP_LIBRARY: another build of the same library (tuning builds with in-kernel timers); default: the in-tree one
LIB_PATH = os.environ.get('AZHIP_LIBRARY') or os.path.join(HERE, 'libazhip.so')

GAME_TTT, GAME_C4, GAME_GENERIC = 0, 1, 2
MAX_ACTIONS, MAX_MOVES = 9, 42
NO_NODE = 0xFFFFFFFF
LEAF_NONE, LEAF_EVAL, LEAF_TERMINAL, LEAF_ROOT, LEAF_CACHED = 0, 1, 2, 3, 4
WINNER_NONE, WINNER_DRAW = -2, -1


class AzError(RuntimeError):
    pass


class EngineConfig(C.Structure):
    _fields_ = [('game', C.c_int32), ('num_games', C.c_int32), ('node_capacity', C.c_int32),
                ('num_simulations', C.c_int32), ('exploration_depth', C.c_int32),
                ('reuse_search_tree', C.c_int32), ('device', C.c_int32), ('model_version', C.c_int32),
                ('alpha', C.c_double), ('epsilon', C.c_double), ('c_base', C.c_double),
                ('c_init', C.c_double), ('seed', C.c_uint64), ('first_game_uid', C.c_uint64),
                ('uid_stride', C.c_uint64)]


class GameRecord(C.Structure):
    _fields_ = [('uid', C.c_uint64), ('num_moves', C.c_int32), ('winner', C.c_int32),
                ('model_version', C.c_int32), ('first_model_version', C.c_int32),
                ('history', C.c_uint8 * (MAX_MOVES + 6)), ('num_edges', C.c_uint8 * (MAX_MOVES + 6)),
                ('visits', (C.c_uint32 * MAX_ACTIONS) * MAX_MOVES)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ('sims', 'select_steps', 'leaf_evals', 'root_evals', 'moves',
                                          'games', 'errors')]


# every symbol include/azhip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    'az_last_error': (C.c_char_p, []),
    'az_version': (C.c_int, []),
    'az_engine_create': (C.c_int, [C.POINTER(EngineConfig), C.POINTER(_P)]),
    'az_engine_destroy': (C.c_int, [_P]),
    'az_reset_games': (C.c_int, [_P, _P]),
    'az_abandon_games': (C.c_int, [_P, _P]),
    'az_set_puct_lut': (C.c_int, [_P, _P, C.c_int32]),
    'az_set_model_version': (C.c_int, [_P, C.c_int32]),
    'az_begin_move': (C.c_int, [_P, _P, _P, _P]),
    'az_apply_root': (C.c_int, [_P, _P, _P, _P]),
    'az_select': (C.c_int, [_P, _P, _P, _P]),
    'az_expand_backup': (C.c_int, [_P, _P, _P, _P]),
    'az_sim_step': (C.c_int, [_P, _P, _P, _P, _P, _P]),
    'az_eval_cache_enable': (C.c_int, [_P, C.c_int32]),
    'az_eval_cache_hits': (C.c_int, [_P, C.POINTER(C.c_uint64), _P]),
    'az_run_sims_uniform': (C.c_int, [_P, C.c_int32, _P]),
    'az_begin_move_uniform': (C.c_int, [_P, _P, _P]),
    'az_choose_and_advance': (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    'az_drain_finished': (C.c_int, [_P, C.POINTER(GameRecord), C.c_int32, C.POINTER(C.c_int32), _P]),
    'az_take_finished_device': (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    'az_get_stats': (C.c_int, [_P, C.POINTER(Stats), _P]),
    'az_engine_error_flags': (C.c_int, [_P]),
    'az_set_position': (C.c_int, [_P, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, _P, C.c_int32, C.c_uint32]),
    'az_get_tree': (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]),
    'az_node_words': (C.c_int, [_P]),
    'az_generic_root': (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P]),
    'az_generic_select': (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), _P, _P]),
    'az_generic_expand_backup': (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_float, _P]),
    'az_positions_from_history': (C.c_int, [C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    'az_game_query': (C.c_int, [C.c_int32, _P, _P, C.c_int32, _P, _P, _P, _P]),
    'az_game_step': (C.c_int, [C.c_int32, _P, _P, _P, C.c_int32, _P, _P]),
    'az_render': (C.c_int, [C.c_int32, _P, C.c_int32, _P, _P, _P]),
    'az_masked_softmax': (C.c_int, [C.c_int32, _P, _P, C.c_int32, _P, _P]),
    'az_ucb_score': (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, _P, _P, _P]),
    'az_c4net_blob_bytes': (C.c_int, []),
    'az_c4net_bf16_forward': (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, _P, _P]),
    'az_c4net_f32_blob_floats': (C.c_int, []),
    'az_c4net_hx_blob_bytes': (C.c_int, []),
    'az_c4net_hx_forward': (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, _P, _P]),
    'az_examples_from_records': (C.c_int, [C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    'az_c4net128_blob_bytes': (C.c_int, []),
    'az_c4net128_workspace_bytes': (C.c_long, [C.c_int32]),
    'az_c4net128_bf16_forward': (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, _P, _P, _P]),
    'az_c4net_f32_forward': (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P, _P]),
    'az_c4_move_fused_max_games': (C.c_int, [_P]),
    'az_c4_move_fused': (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P, _P]),
    'az_c4_move_fused_hx': (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P, _P]),
    'az_c4_move_fused_limit_workgroups': (C.c_int, [_P, C.c_int32]),
    'az_c4_move_fused_counters': (C.c_int, [_P, C.POINTER(C.c_uint64), _P]),
    'az_c4_move_fused_static': (C.c_int, [_P, _P, C.c_int32, _P]),
    'az_loss_fwd_bwd': (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, C.c_int32, _P, _P, _P, _P]),
    'az_sgd_momentum': (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int32, _P]),
    'az_sgd_momentum_lr_dev': (C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_float, C.c_float, _P]),
    'az_sgd_momentum_guarded': (C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_float, C.c_float, _P, C.c_float, _P, _P]),
    'az_sample_window': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_uint64, _P, C.c_int32, _P, _P, _P, _P, _P]),
    'az_c4_train_workspace_bytes': (C.c_long, [C.c_int32, C.c_int32]),
    'az_c4_train_fwd_bwd': (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, _P, _P, _P, _P, _P]),
    'az_c4net_pack_bf16': (C.c_int, [_P, C.c_int32, C.c_double, _P, _P]),
    'az_c4net_pack_hx': (C.c_int, [_P, C.c_int32, C.c_double, _P, _P, _P]),
    'az_c4_train_fwd_bwd_chain': (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, _P, _P, _P, _P, _P]),
    'az_c4_train_one_launch_max_batch': (C.c_int, [C.c_int32, _P]),
    'az_tttnet_f32_blob_floats': (C.c_int, [C.c_int32]),
    'az_tttnet_f32_forward': (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P, _P, _P]),
}

_lib = None


def load():
    """Load libazhip.so; raises AzError when it is not built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AzError(f'{LIB_PATH} not found: build it with `make -C alphazero_amd/csrc` '
                      '(python -c "import __graft_entry__ as g; g.build()")')
    # The library is meant to sit next to PyTorch-ROCm in one process and to share its HIP runtime
    # (streams and device pointers cross the boundary): torch is loaded first, so that the
    # libamdhip64 it ships is the one both bind to.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise AzError(load().az_last_error().decode())


def ptr(t):
    """Device (or host) pointer of a torch tensor / None."""
    return None if t is None else C.c_void_p(t.data_ptr())
