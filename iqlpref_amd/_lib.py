"""ctypes binding of libiqlhip.so (include/iqlhip.h).

There is no CPU fallback: importing this module without the built library, or
calling into it without a ROCm device, raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# IQLHIP_LIB: load another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("IQLHIP_LIB") or os.path.join(HERE, "libiqlhip.so")

PREC_FP32 = 0
PREC_BF16 = 1
MAX_CRITICS = 8
MAX_HIDDEN = 6
N_TENSORS = 2 * (MAX_HIDDEN + 1) * (MAX_CRITICS + 2) + 1
MLP_MAX_LAYERS = 8
MAX_GROUP = 16
ACT_FLAX_BASE = 8  # iqlhip_mlp_desc activation code 8 + i = entry i of reward_models/q_mlp.py:121-130
ABI_VERSION = 6

ERR_INVALID = -1
ERR_HIP = -2
ERR_UNSUPPORTED = -3
ERR_NOMEM = -4


class ReplayView(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("n_rows", C.c_int64), ("row_stride", C.c_int32),
                ("state_dim", C.c_int32), ("action_dim", C.c_int32), ("generation", C.c_uint64)]


class TrainerConfig(C.Structure):
    _fields_ = [("state_dim", C.c_int32), ("action_dim", C.c_int32),
                ("hidden_dim", C.c_int32), ("batch_size", C.c_int32),
                ("deterministic", C.c_int32), ("precision", C.c_int32),
                ("dropout_p", C.c_float),
                ("discount", C.c_float), ("tau", C.c_float), ("beta", C.c_float),
                ("iql_tau", C.c_float),
                ("lr_q", C.c_double), ("lr_v", C.c_double), ("lr_actor", C.c_double),
                ("adam_beta1", C.c_double), ("adam_beta2", C.c_double),
                ("adam_eps", C.c_double),
                ("cosine_t_max", C.c_int64), ("seed", C.c_uint64),
                ("n_critics", C.c_int32), ("polyak_form", C.c_int32), ("n_hidden", C.c_int32)]


class Arenas(C.Structure):
    _fields_ = [("params", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("target", C.c_void_p), ("grads", C.c_void_p)]


class MlpDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("dims", C.c_int32 * (MLP_MAX_LAYERS + 1)),
                ("weights", C.c_void_p * MLP_MAX_LAYERS),
                ("biases", C.c_void_p * MLP_MAX_LAYERS),
                ("w_in_out", C.c_int32), ("hidden_act", C.c_int32), ("out_act", C.c_int32),
                ("dropout_p", C.c_float), ("dropout_call", C.c_uint32), ("dropout_seed", C.c_uint64)]


class PtWeights(C.Structure):
    _fields_ = ([("state_dim", C.c_int32), ("action_dim", C.c_int32), ("embd_dim", C.c_int32),
                 ("num_heads", C.c_int32), ("inter_dim", C.c_int32), ("num_layers", C.c_int32),
                 ("n_temb", C.c_int32), ("eps", C.c_float)] +
                [(n, C.c_void_p) for n in (
                    "state_wT", "state_b", "action_wT", "action_b", "temb", "sln_w", "sln_b",
                    "ln0_w", "ln0_b", "qkv_w", "qkv_b", "attn_out_w", "attn_out_b",
                    "ln1_w", "ln1_b", "mlp_in_w", "mlp_in_b", "mlp_out_w", "mlp_out_b",
                    "lnf_w", "lnf_b", "pref_w_last")] +
                [("pref_b_last", C.c_float)])


# every symbol include/iqlhip.h declares: name -> (restype, argtypes)
P = C.c_void_p
SYMBOLS = {
    "iqlhip_last_error": (C.c_char_p, []),
    "iqlhip_abi_version": (C.c_int, []),
    "iqlhip_build_tag": (C.c_char_p, []),
    "iqlhip_replay_row_stride": (C.c_int32, [C.c_int32, C.c_int32]),
    "iqlhip_replay_next_offset": (C.c_int32, [C.c_int32, C.c_int32]),
    "iqlhip_replay_pack": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                     P, P, P, P, P, P]),
    "iqlhip_replay_pack_normalized": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                                P, P, P, P, P, P, P, P]),
    "iqlhip_prep_keep_mask": (C.c_int, [P, P, C.c_int64, C.c_int32, C.c_int32, P, P, P]),
    "iqlhip_prep_reward_range": (C.c_int, [P, P, C.c_int64, C.c_int32, P, C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), P]),
    "iqlhip_prep_modify_reward": (C.c_int, [P, C.c_int64, P, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                            C.c_double, C.c_int32, P]),
    "iqlhip_prep_state_stats": (C.c_int, [P, C.c_int64, C.c_int32, C.c_double, P, P, P]),
    "iqlhip_replay_sample": (C.c_int, [C.POINTER(ReplayView), C.c_int32, P, C.c_uint64,
                                       C.c_uint64, P, P, P, P, P, P, P]),
    "iqlhip_arena_layout": (C.c_int, [C.POINTER(TrainerConfig), C.POINTER(C.c_int64 * N_TENSORS),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "iqlhip_trainer_create": (C.c_int, [C.POINTER(P), C.POINTER(TrainerConfig), C.POINTER(Arenas)]),
    "iqlhip_trainer_destroy": (C.c_int, [P]),
    "iqlhip_trainer_step_kind": (C.c_int, [P, C.POINTER(C.c_int32)]),
    "iqlhip_trainer_sync_weights": (C.c_int, [P, P]),
    "iqlhip_trainer_set_step": (C.c_int, [P, C.c_int64]),
    "iqlhip_trainer_get_step": (C.c_int, [P, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "iqlhip_trainer_set_lr": (C.c_int, [P, C.c_double, C.c_double, C.c_double]),
    "iqlhip_train_steps": (C.c_int, [P, C.POINTER(ReplayView), C.c_int64, P, P, P, C.c_int32, P]),
    "iqlhip_group_create": (C.c_int, [C.POINTER(P), C.POINTER(P), C.c_int32]),
    "iqlhip_group_destroy": (C.c_int, [P]),
    "iqlhip_group_train_steps": (C.c_int, [P, C.POINTER(ReplayView), C.c_int64, C.POINTER(P), C.POINTER(P),
                                           C.POINTER(P), C.c_int32, P]),
    "iqlhip_group_set_timing": (C.c_int, [P, C.c_int32]),
    "iqlhip_group_get_timing": (C.c_int, [P, C.POINTER(C.c_double * 3), C.POINTER(C.c_int64)]),
    "iqlhip_stream_create_cu_slice": (C.c_int, [C.POINTER(P), C.c_int32, C.c_int32]),
    "iqlhip_stream_destroy": (C.c_int, [P]),
    "iqlhip_train_batch": (C.c_int, [P, P, P, P, P, P, P, P, P]),
    "iqlhip_forward": (C.c_int, [P, C.c_int32, P, P, C.c_int64, P, P]),
    "iqlhip_mlp_forward": (C.c_int, [C.POINTER(MlpDesc), P, C.c_int64, C.c_int32, P, C.c_int32, P]),
    "iqlhip_cvar_tail_mean": (C.c_int, [P, C.c_int32, C.c_int64, C.c_int32, P, P]),
    "iqlhip_pt_relabel": (C.c_int, [C.POINTER(PtWeights), P, P, C.c_int64, P, P, P, C.c_int64, C.c_int32,
                                    P, P]),
    "iqlhip_step_cost": (C.c_int, [C.POINTER(TrainerConfig), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double)]),
    "iqlhip_trainer_set_timing": (C.c_int, [P, C.c_int32]),
    "iqlhip_trainer_get_timing": (C.c_int, [P, C.POINTER(C.c_double * 3), C.POINTER(C.c_int64)]),
}

_lib = None


def load():
    """Load libiqlhip.so; raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m iqlpref_amd.build` "
            "(hipcc --offload-arch=gfx950). iqlpref_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.iqlhip_abi_version() != ABI_VERSION:
        raise ImportError("libiqlhip.so ABI version mismatch; rebuild it")
    if not os.environ.get("IQLHIP_LIB"):  # an explicitly named other build is taken as it is
        from . import build as _build
        want, have = _build.source_tag(), lib.iqlhip_build_tag().decode()
        if want != have:
            raise ImportError(
                f"{LIB_PATH} was built from other sources (library tag {have}, sources {want}): "
                "rebuild it with `python -m iqlpref_amd.build`")
    _lib = lib
    return lib


def build_tag():
    """Source hash the loaded library was built from (iqlpref_amd/build.py)."""
    return load().iqlhip_build_tag().decode()


def check(rc):
    """Map a negative iqlhip_status to the exception type the reference raises."""
    if rc == 0:
        return
    msg = load().iqlhip_last_error().decode("utf-8", "replace")
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def require_gpu(device):
    import torch
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError(
            f"iqlpref_amd runs on a ROCm GPU only (got device={device!r}, "
            f"torch.cuda.is_available()={torch.cuda.is_available()}); there is no CPU path")
    return dev


def stream_ptr():
    """The current HIP stream of the current device as a void* (what every entry point launches on)."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:  # (no Stream object: ~0.3 us instead of ~3 us on the launch path)
        return C.c_void_p(raw(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
