"""ctypes binding of include/xq_capi.h (libxqhip.so).  Plumbing only — no compute happens in Python.

The library is built in-tree by `make -C cn_chess_ai_amd/csrc` (see __graft_entry__.build()).  There is no CPU
fallback: if the shared object is missing or no HIP device is usable, calls raise XqError.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# XQ_LIBXQHIP: another build of the same library (same-box A/B of two builds, tools/); never a different implementation
LIB_PATH = os.environ.get("XQ_LIBXQHIP") or os.path.join(HERE, "libxqhip.so")

XQ_OK = 0
MAX_MOVES = 128
BOARD_WORDS = 12
MAX_LAYERS = 8
NET_ONLINE, NET_TARGET = 0, 1
BACKPROP_REFERENCE, BACKPROP_TEXTBOOK = 0, 1
TD_ONLINE_NET, TD_TARGET_NET, TD_DOUBLE = 0, 1, 2
PRECISION_F32, PRECISION_BF16, PRECISION_BF16_FULL = 0, 1, 2
QMAX_FULL, QMAX_SCREENED = 0, 1
ORDER_RING_CONTENTS, ORDER_RING_PRIORITIES, ORDER_RING_DRAW, ORDER_TRAINER_PARAMS, ORDER_ALL = 1, 2, 4, 8, 15

STATUS_NAMES = {1: "XQ_ERR_INVALID_ARGUMENT", 2: "XQ_ERR_RUNTIME", 3: "XQ_ERR_NO_DEVICE", 4: "XQ_ERR_IO",
                5: "XQ_ERR_UNDEFINED_UPSTREAM"}


class XqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


class StepResult(C.Structure):
    _fields_ = [("action", C.c_int32), ("n_moves", C.c_int32), ("reward", C.c_int32),
                ("captured", C.c_uint8), ("valid", C.c_uint8), ("done", C.c_uint8), ("terminated", C.c_uint8),
                ("winner", C.c_uint8), ("explored", C.c_uint8), ("move_count", C.c_uint16),
                ("red_score", C.c_int16), ("black_score", C.c_int16)]


class EpisodeRecord(C.Structure):
    _fields_ = [("game_id", C.c_uint32), ("episode", C.c_uint32), ("red_score", C.c_int16),
                ("black_score", C.c_int16), ("move_count", C.c_uint16), ("winner", C.c_uint8), ("reserved", C.c_uint8)]


class KernelSpan(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("start_ms", C.c_float), ("end_ms", C.c_float)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_float), ("launches", C.c_int), ("flops", C.c_double),
                ("bytes", C.c_double), ("exact_launches", C.c_int), ("reserved", C.c_int)]


class TrainerConfig(C.Structure):
    _fields_ = [("n_games", C.c_int), ("layer_sizes", C.c_int * (MAX_LAYERS + 1)), ("n_sizes", C.c_int),
                ("learning_rate", C.c_double), ("gamma", C.c_double), ("epsilon", C.c_double),
                ("replay_capacity", C.c_int), ("minibatch", C.c_int), ("td_net", C.c_int), ("backprop_mode", C.c_int),
                ("target_sync_interval", C.c_int), ("mean_gradient", C.c_int), ("seed", C.c_uint64),
                ("first_game_id", C.c_uint32), ("collects_per_update", C.c_int), ("overlap_collect", C.c_int),
                ("prioritized", C.c_int), ("per_alpha", C.c_double), ("per_beta", C.c_double), ("per_eps", C.c_double),
                ("precision", C.c_int)]


assert C.sizeof(StepResult) == 24 and C.sizeof(EpisodeRecord) == 16

_vp, _i, _u32, _u64, _d = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_double
_pi, _pd, _pf = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_float)
_pu8, _pu16, _pu64 = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_uint64)
_pvp = C.POINTER(C.c_void_p)

# every symbol include/xq_capi.h declares: name -> argtypes (restype is int unless listed in _RESTYPES)
PROTOTYPES = {
    "xq_last_error": [],
    "xq_version": [],
    "xq_device_count": [_pi],
    "xq_set_device": [_i],
    "xq_stream_synchronize": [_vp],
    "xq_stream_wait_stream": [_vp, _vp],
    "xq_stream_create": [_i, _i, _pvp],
    "xq_stream_destroy": [_vp],
    "xq_debug_stream_delay": [_vp, _i],
    "xq_debug_stream_gate": [_vp, _i, _pvp],
    "xq_debug_gate_release": [_vp],
    "xq_debug_gate_destroy": [_vp],
    "xq_stream_query": [_vp, _pi],
    "xq_debug_set_stream_ordering": [C.c_uint],
    "xq_env_stream": [_vp, _pvp],
    "xq_replay_stream": [_vp, _pvp],
    "xq_dqn_stream": [_vp, _pvp],
    "xq_event_create": [_pvp],
    "xq_event_destroy": [_vp],
    "xq_event_record": [_vp, _vp],
    "xq_event_elapsed_ms": [_vp, _vp, _pf],
    "xq_env_create": [_i, _u64, _u32, _vp, _pvp],
    "xq_env_destroy": [_vp],
    "xq_env_num_games": [_vp, _pi],
    "xq_env_reset": [_vp],
    "xq_env_set_state": [_vp, _i, _i, _pu8, _pi],
    "xq_env_get_state": [_vp, _i, _i, _pu8, _pi],
    "xq_env_legal_moves": [_vp, _i, _pu16, _pi],
    "xq_env_legal_moves_dev": [_vp, _i, _vp, _vp],
    "xq_env_valid_matrix": [_vp, _i, _pu8],
    "xq_env_get_winner": [_vp, _i, _i, _pu8],
    "xq_env_rule_matrix": [_vp, _i, _pu8],
    "xq_env_rule_query": [_vp, _i, _i, _i, _i, _i, _i, _pi],
    "xq_env_step": [_vp, _pi, _i, C.POINTER(StepResult)],
    "xq_env_selfplay_step": [_vp, _vp, _i, _u32, _vp, _vp],
    "xq_env_selfplay_step_host": [_vp, _pf, _u32, C.POINTER(StepResult)],
    "xq_env_drain_episodes": [_vp, C.POINTER(EpisodeRecord), _i, _pi, _pu64],
    "xq_env_counters": [_vp, _pu64],
    "xq_env_boards_dev": [_vp],
    "xq_env_meta_dev": [_vp],
    "xq_replay_create": [_i, _u64, _vp, _pvp],
    "xq_replay_destroy": [_vp],
    "xq_replay_size": [_vp, _pi, _pi, _pu64],
    "xq_replay_push_host": [_vp, _i, _pu8, _pi, _pf, _pu8, _pu8],
    "xq_replay_sample": [_vp, _i, _pi],
    "xq_replay_sample_window": [_vp, _i, _i, _i, _pi],
    "xq_replay_get": [_vp, _i, _pu8, _pi, _pf, _pu8, _pu8],
    "xq_replay_enable_per": [_vp, _d, _d, _d],
    "xq_replay_per_rebuild": [_vp, _i, _i],
    "xq_replay_sample_prioritized": [_vp, _i, _pi, _pf],
    "xq_replay_set_priorities": [_vp, _i, _i, _pf],
    "xq_replay_get_priorities": [_vp, _i, _i, _pf],
    "xq_replay_per_stats": [_vp, _pf, _pf, _pi],
    "xq_dqn_create": [_pi, _i, _d, _d, _u64, _vp, _pvp],
    "xq_dqn_destroy": [_vp],
    "xq_dqn_set_precision": [_vp, _i],
    "xq_dqn_set_qmax_mode": [_vp, _i],
    "xq_dqn_set_l0_derive": [_vp, _i],
    "xq_dqn_set_td_tail": [_vp, _i],
    "xq_dqn_set_l0_grad_mode": [_vp, _i],
    "xq_dqn_set_refine_stage": [_vp, _i],
    "xq_dqn_set_exchange_overlap": [_vp, _i],
    "xq_dqn_calibrate_exchange": [_vp, _d, _pd, _pi],
    "xq_dqn_exchange_calibration": [_vp, _pi, _pd, _pd, _pi],
    "xq_dqn_qmax_stats": [_vp, C.POINTER(C.c_uint64)],
    "xq_dqn_qmax_guard": [_vp, _pu64, _pi],
    "xq_dqn_num_params": [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
    "xq_dqn_set_params": [_vp, _i, _pd, _pd],
    "xq_dqn_get_params": [_vp, _i, _pd, _pd],
    "xq_dqn_forward": [_vp, _i, _pd, _i, _pd],
    "xq_dqn_forward_boards_dev": [_vp, _i, _vp, _i, _i, _vp, _i],
    "xq_dqn_select_q_dev": [_vp, _vp, _i, _vp],
    "xq_dqn_backpropagate": [_vp, _pd, _pd, _i, _d, _d, _i],
    "xq_dqn_update_target": [_vp],
    "xq_dqn_save_model": [_vp, C.c_char_p],
    "xq_dqn_load_model": [_vp, C.c_char_p],
    "xq_dqn_td_grads": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "xq_dqn_apply_grads": [_vp, _d, _d],
    "xq_dqn_grad_buffer": [_vp, _pvp, C.POINTER(C.c_size_t)],
    "xq_dqn_td_grads_replay": [_vp, _vp, _i, _i, _i],
    "xq_dqn_td_update_host": [_vp, _i, _pu8, _pu8, _pi, _pf, _pu8, _i, _i, _d, _d, _pf, _pf],
    "xq_dqn_last_loss": [_vp, _pd],
    "xq_dqn_last_td_values": [_vp, _i, _pf, _pf],
    "xq_dqn_kernel_stats": [_vp, _i, C.POINTER(KernelStat), _i, _pi],
    "xq_dqn_set_fused_apply": [_vp, _i],
    "xq_dqn_kernel_filter": [_vp, C.c_char_p],
    "xq_dqn_kernel_timeline": [_vp, C.POINTER(KernelSpan), _i, _pi],
    "xq_comm_unique_id": [_pu8],
    "xq_comm_create": [_i, _i, _pu8, _pvp],
    "xq_comm_create_from_file": [_i, _i, C.c_char_p, _d, _pvp],
    "xq_comm_destroy": [_vp],
    "xq_comm_info": [_vp, _pi, _pi, _pu64, _pu64],
    "xq_comm_allreduce": [_vp, _vp, C.c_size_t, _vp],
    "xq_comm_sum_u64": [_vp, _pu64],
    "xq_dqn_set_comm": [_vp, _vp],
    "xq_allreduce_grads": [_vp, _vp],
    "xq_trainer_set_comm": [_vp, _vp],
    "xq_trainer_create": [C.POINTER(TrainerConfig), _vp, _pvp],
    "xq_trainer_destroy": [_vp],
    "xq_trainer_env": [_vp, _pvp],
    "xq_trainer_dqn": [_vp, _pvp],
    "xq_trainer_replay": [_vp, _pvp],
    "xq_trainer_random_plies": [_vp, _i],
    "xq_trainer_set_td_net": [_vp, _i],
    "xq_trainer_collect": [_vp],
    "xq_trainer_learn_grads": [_vp],
    "xq_trainer_learn_apply": [_vp, _i],
    "xq_trainer_step": [_vp, _i],
    "xq_trainer_counters": [_vp, _pu64, _pu64, _pu64],
}
_RESTYPES = {"xq_last_error": C.c_char_p, "xq_env_boards_dev": C.c_void_p, "xq_env_meta_dev": C.c_void_p}

_lib = None


def load():
    """Loads libxqhip.so; raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XqError(3, f"{LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "or `make -C cn_chess_ai_amd/csrc`; there is no CPU fallback")
    # torch bundles its own libamdhip64.so.7; importing it FIRST makes libxqhip bind to that same runtime instance
    # (one HIP runtime per process: two instances cannot both own the GPU, and torch.distributed/RCCL must see the
    # very allocations this library makes).  Without torch the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def check(rc):
    if rc != XQ_OK:
        msg = load().xq_last_error()
        raise XqError(rc, msg.decode() if msg else "")
    return rc


def call(name, *args):
    return check(getattr(load(), name)(*args))


def device_count():
    n = C.c_int32(0)
    lib = load()
    rc = lib.xq_device_count(C.byref(n))
    return n.value if rc == XQ_OK else 0
