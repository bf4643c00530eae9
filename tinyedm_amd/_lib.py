"""ctypes binding of the C-ABI HIP library (include/tinyedm_hip.h).

There is NO CPU fallback: if the shared library is missing or a kernel reports an error the
call raises.  Only raw device pointers, sizes and a hipStream_t cross this boundary.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EDM_LIB_PATH") or os.path.join(_HERE, "libtinyedm_hip.so")   # override: A/B runs in tools/

P, I, L, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
U, U64 = ctypes.c_uint, ctypes.c_ulonglong

# name -> argtypes (every function returns int status unless listed in _RET)
SIGNATURES = {
    "edm_version": [],
    "edm_last_error": [],
    # runtime.hip
    "edm_init": [I],
    "edm_graph_replay_safe": [],
    # elementwise.hip
    "edm_pixelnorm_silu_fwd": [P, P, P, P, L, I, P],
    "edm_pixelnorm_silu_bwd": [P, P, P, F, P, P, P, L, I, P],
    "edm_pool_pixelnorm_silu_fwd": [P, P, P, P, I, I, I, I, P],
    "edm_pool_pixelnorm_silu_bwd": [P, P, P, F, P, P, P, I, I, I, I, P],
    "edm_up2_silu": [P, P, P, I, I, I, I, P],
    "edm_silu_fwd": [P, P, L, P],
    "edm_silu_bwd": [P, P, P, F, P, L, P],
    "edm_axpby": [P, F, P, F, P, L, P],
    "edm_mod_silu_drop_fwd": [P, P, L, P, P, I, I, I, F, U64, U, U, P, P],
    "edm_mod_silu_drop_bwd": [P, P, L, P, P, P, P, P, L, P, I, I, I, F, U64, U, U, P, P],
    "edm_mod_silu_drop_bwd_raw": [P, P, L, P, P, P, P, L, I, I, I, F, U64, U, U, P, P],
    "edm_dropout_mask": [P, L, F, U64, U, U, P],
    "edm_pool2": [P, P, I, I, I, I, F, P],
    "edm_up2": [P, P, P, I, I, I, I, F, P],
    "edm_reduce_hw": [P, L, P, L, P, I, I, I, F, P],
    "edm_scalelong_fwd": [P, P, P, P, P, I, I, I, P],
    "edm_scalelong_bwd": [P, P, P, P, P, P, P, P, P, I, I, I, P],
    "edm_skip_gate_fwd": [P, P, P, P, P, P, I, I, I, I, P],
    "edm_skip_gate_bwd": [P, L, I, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    "edm_skip_gate_wgrad_multi": [P, I, P, P, I, P],
    "edm_skip_gate_wgrad_multi_table_bytes": [],
    "edm_skip_gate_fwd_multi": [P, I, P, P, I, P],
    "edm_skip_gate_fwd_multi_table_bytes": [],
    "edm_skip_gate_bwd_multi": [P, I, P, P, I, P],
    "edm_skip_half_bwd_multi": [P, I, P, P, I, P],
    "edm_skip_gate_bwd_multi_table_bytes": [],
    "edm_concat_gate_fwd": [P, P, P, P, P, I, I, I, I, P],
    "edm_concat_gate_bwd": [P, P, P, P, P, I, I, I, I, P],
    "edm_skip_half_fwd": [P, P, P, P, I, I, I, I, P],
    "edm_skip_half_bwd": [P, P, P, P, I, I, I, P],
    "edm_precond_in": [P, P, I, F, P, I, I, I, I, P],
    "edm_conv_out_fwd": [P, P, P, P, P, I, F, P, P, I, I, I, I, P],
    "edm_conv_out_bwd": [P, P, P, P, P, P, I, F, P, P, P, I, I, I, I, P],
    "edm_nchw_to_nhwc_bf16": [P, P, I, I, I, P],
    "edm_nhwc_bf16_to_nchw": [P, P, I, I, I, P],
    # conv_igemm.hip / conv_wgrad.hip
    "edm_conv_igemm": [P, P, P, P, F, F, I, I, I, I, I, I, P],
    "edm_conv_igemm_v2": [P, P, P, P, F, F, I, I, I, I, I, I, P],
    "edm_conv_igemm_o": [P, P, P, L, P, P, L, I, P, F, F, I, I, I, I, I, I, I, I, P],
    "edm_conv3x3_fold_supported": [I, I, I, I, I, I],
    "edm_conv3x3_fold": [P, P, P, L, P, I, P, L, P, F, F, I, I, I, I, I, P],
    "edm_conv_igemm_v6": [P, P, P, P, F, F, I, I, I, I, I, I, P],
    "edm_conv_igemm_s": [P, P, P, P, F, F, I, I, I, I, I, I, P],
    "edm_conv3x3_mod": [P, P, P, P, P, L, P, F, U64, U, U, I, I, I, I, I, I, P, I, P],
    "edm_conv3x3_modbwd": [P, P, F, P, P, L, P, P, P, L, F, U64, U, U, I, I, I, I, I, I, P, I, P],
    "edm_mod_finish_multi": [P, P, P, L, P, I, I, P],
    "edm_conv3x3_silubwd": [P, P, P, P, F, P, I, I, I, I, I, I, P],
    "edm_mod_finish": [P, P, L, P, P, L, P, I, I, P],
    "edm_conv_wgrad_nsplit": [I, I, I, I, I, I],
    "edm_conv_wgrad_v2": [P, P, P, I, I, I, I, I, I, I, P],
    "edm_conv_wgrad_1x1_nsplit": [L, I, I],
    "edm_conv_wgrad_1x1_nsplit_grouped": [L, I, I],
    "edm_conv_wgrad_1x1": [P, P, P, L, I, I, I, P],
    "edm_conv_wgrad_1x1_group": [P, I, P, P, I, P],
    "edm_conv_wgrad_1x1_group_table_bytes": [],
    # conv_wgrad3.hip (items = host array of WGrad3Item)
    "edm_wgrad3_workspace": [P, I],
    "edm_wgrad3_group": [P, I, P, L, P, P, I, P],
    "edm_wgrad3_table_bytes": [],
    "edm_wgrad3_max_layers": [],
    # attention.hip
    "edm_attention_fwd": [P, P, I, I, I, I, P],
    "edm_attention_bwd": [P, P, P, P, I, I, I, I, P],
    # attention_fused.hip
    "edm_attention_qkv_supported": [I, I, I],
    "edm_attention_qkv_fwd": [P, P, P, P, I, I, I, I, I, P],
    "edm_attention_qkv_bwd": [P, P, P, P, P, P, P, F, I, I, I, I, I, P],
    # linear.hip
    "edm_linear_fwd": [P, P, P, I, I, I, P],
    "edm_linear_dgrad": [P, P, P, I, I, I, I, P],
    "edm_linear_wgrad": [P, P, P, I, I, I, I, P],
    "edm_fourier_fwd": [P, I, P, P, P, I, I, P],
    "edm_embed_combine_fwd": [P, P, P, F, I, P, P, I, I, P],
    "edm_embed_combine_bwd": [P, P, P, F, I, P, P, I, I, P],
    # optim.hip
    "edm_diffuse": [P, P, P, F, F, I, L, U64, U, P, P],
    "edm_diffuse_given": [P, P, P, P, P, F, F, I, L, P],
    "edm_weighted_mse": [P, P, P, P, F, P, P, I, L, P, P, P],
    "edm_adam_ema": [P, P, P, P, P, L, F, F, F, F, I, F, F, P, I, P, P],
    "edm_heun_euler": [P, P, F, F, P, P, L, P, P],
    "edm_heun_correct": [P, P, P, P, F, F, P, L, P, P],
    "edm_scale_f32": [P, F, P, L, P],
    # weights.hip
    "edm_weight_prep": [P, I, I, I, I, P, P, P, P, I, P],
    "edm_weight_prep_multi": [P, P, I, I, I, P],
    "edm_wgrad_finish": [P, I, P, P, P, I, I, I, I, F, I, P],
    "edm_wgrad_finish_multi": [P, I, P, P, I, P],
    "edm_wgrad_finish_multi_table_bytes": [],
    # eval_f32.hip (reference-precision evaluation path)
    "edm_f32_conv": [P, P, P, P, F, F, P, L, P, I, I, I, I, I, I, I, P],
    "edm_f32_attention": [P, P, I, I, I, I, P],
    "edm_split_attention": [P, P, P, I, I, I, I, P],
    "edm_f32_to_pairs": [P, P, L, I, P],
    "edm_split_pack": [P, P, I, I, I, I, P],
    "edm_split_conv": [P, P, P, P, P, F, F, P, L, P, I, I, I, I, I, I, P],
    "edm_split_conv_o": [P, P, P, P, L, L, P, P, F, F, P, L, P, I, I, I, I, I, I, P],
    "edm_split_conv_fold": [P, P, P, P, I, P, P, L, L, P, F, F, I, I, I, I, I, P],
    "edm_f32_skip_half": [P, P, P, P, I, I, I, I, P],
    "edm_f32_pool_pixelnorm_silu": [P, P, P, I, I, I, I, I, P],
    "edm_f32_up2_silu": [P, P, P, I, I, I, I, I, P],
    "edm_f32_pixelnorm_silu": [P, P, P, L, I, I, P],
    "edm_f32_silu": [P, P, L, I, P],
    "edm_f32_pool2": [P, P, I, I, I, I, P],
    "edm_f32_up2": [P, P, I, I, I, I, P],
    "edm_f32_skip_gate": [P, P, P, P, I, I, I, I, P],
    "edm_f32_concat_gate": [P, P, P, P, P, I, I, I, I, I, P],
    "edm_f32_precond_in": [P, P, I, F, P, I, I, I, I, P],
    "edm_f32_conv_out": [P, P, P, P, P, I, F, P, I, I, I, I, P],
    "edm_f32_nchw_to_nhwc": [P, P, I, I, I, P],
    "edm_f32_nhwc_to_nchw": [P, P, I, I, I, P],
    # data.hip
    "edm_u8_gather_normalize": [P, P, P, I, I, I, I, L, F, F, I, U64, U, P],
    "edm_denormalize_u8": [P, P, L, F, F, P],
    "edm_prediction_to_u8_nhwc": [P, P, I, I, I, I, P, P, P],
}
# include/tinyedm_hip_diag.h: tools-only entry points, bound on demand by call()
DIAG_SIGNATURES = {
    "edm_conv_igemm_v2_stamp": [P, P, P, I, I, I, I, I, P, P],
    "edm_conv_igemm_v2_ablate": [P, P, P, I, I, I, I, I, I, P],
    "edm_wgrad3_probe": [P, P],
    "edm_wgrad3_plan_ksplit": [P, I, P],
    "edm_v6_persistent_launches": [],
}
_RET = {"edm_last_error": ctypes.c_char_p, "edm_v6_persistent_launches": ctypes.c_long, "edm_wgrad3_workspace": ctypes.c_long, "edm_wgrad3_table_bytes": ctypes.c_long,
        "edm_skip_gate_wgrad_multi_table_bytes": ctypes.c_long, "edm_skip_gate_fwd_multi_table_bytes": ctypes.c_long, "edm_skip_gate_bwd_multi_table_bytes": ctypes.c_long,
        "edm_conv_wgrad_1x1_group_table_bytes": ctypes.c_long, "edm_wgrad_finish_multi_table_bytes": ctypes.c_long}
_NO_STATUS = {"edm_skip_gate_bwd_multi_table_bytes", "edm_skip_gate_fwd_multi_table_bytes", "edm_v6_persistent_launches", "edm_conv3x3_fold_supported", "edm_skip_gate_wgrad_multi_table_bytes", "edm_version", "edm_graph_replay_safe", "edm_last_error", "edm_conv_wgrad_nsplit", "edm_conv_wgrad_1x1_nsplit", "edm_conv_wgrad_1x1_nsplit_grouped", "edm_wgrad3_workspace", "edm_wgrad3_table_bytes", "edm_wgrad3_max_layers",
              "edm_conv_wgrad_1x1_group_table_bytes", "edm_wgrad_finish_multi_table_bytes", "edm_attention_qkv_supported"}

_lib = None


class HipKernelError(RuntimeError):
    pass


class FinishItem(ctypes.Structure):
    """edm_finish_item (include/tinyedm_hip.h)"""
    _fields_ = [("slabs", P), ("w", P), ("grad", P), ("perm", P), ("S", I), ("O", I), ("I", I), ("Ipad", I), ("taps", I),
                ("scale", F), ("accumulate", I)]


class SkipGateWgradItem(ctypes.Structure):
    """edm_skip_gate_wgrad_item (include/tinyedm_hip.h): one ScaleLong gate of a grouped weight-gradient launch."""
    _fields_ = [("ws", P), ("mean", P), ("gW1h", P), ("gW2h", P), ("B", I), ("C", I), ("R", I), ("pad", I)]


class SkipGateFwdItem(ctypes.Structure):
    """edm_skip_gate_fwd_item (include/tinyedm_hip.h): one ScaleLong gate of a grouped forward launch."""
    _fields_ = [("skip", P), ("W1h", P), ("W2h", P), ("mean", P), ("gate", P), ("z1save", P), ("cat", P), ("silu_out", P),
                ("B", I), ("HW", I), ("C", I), ("R", I), ("Ci", I), ("pad", I)]


class SkipGateBwdItem(ctypes.Structure):
    """edm_skip_gate_bwd_item (include/tinyedm_hip.h)"""
    _fields_ = [("gcat", P), ("gcat_stride", L), ("skip", P), ("W1h", P), ("W2h", P), ("gate", P), ("z1save", P), ("gmean", P),
                ("ws", P), ("gskip", P), ("c_off", I), ("B", I), ("HW", I), ("C", I), ("R", I), ("pad", I)]


class SkipHalfBwdItem(ctypes.Structure):
    """edm_skip_half_bwd_item (include/tinyedm_hip.h)"""
    _fields_ = [("gcs", P), ("gate", P), ("gmean", P), ("gskip", P), ("B", I), ("HW", I), ("Cs", I), ("pad", I)]


class WGrad1Item(ctypes.Structure):
    """edm_wgrad1_item (include/tinyedm_hip.h): one 1x1 layer of a grouped weight-gradient launch."""
    _fields_ = [("X", P), ("dY", P), ("slabs", P), ("npix", L), ("Cin", I), ("Cout", I), ("nsplit", I), ("pad", I)]


class WGrad3Item(ctypes.Structure):
    """edm_wgrad3_item (include/tinyedm_hip.h): one 3x3 layer of a grouped weight-gradient launch."""
    _fields_ = [("X", P), ("dY", P), ("w", P), ("grad", P), ("perm", P), ("B", I), ("H", I), ("W", I), ("Cin", I),
                ("Cout", I), ("I", I), ("scale", F), ("accumulate", I)]


def lib() -> ctypes.CDLL:
    """Load (once) and return the HIP library; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"tinyedm_amd: {LIB_PATH} is missing -- build it with `python -m tinyedm_amd.build` "
                "(there is no CPU fallback for the HIP hot path)")
        h = ctypes.CDLL(LIB_PATH)
        for name, args in {**SIGNATURES, **DIAG_SIGNATURES}.items():
            fn = getattr(h, name)  # AttributeError here = header/library out of sync
            fn.argtypes = args
            fn.restype = _RET.get(name, ctypes.c_int)
        _lib = h
    return _lib


N_CALLS = 0     # entry-point invocations so far (bench.py reports launches per step from it)


def call(name: str, *args):
    """Invoke a status-returning entry point; raise HipKernelError on a non-zero status."""
    global N_CALLS
    N_CALLS += 1
    h = lib()
    rc = getattr(h, name)(*args)
    if name in _NO_STATUS:
        return rc
    if rc != 0:
        msg = h.edm_last_error()
        raise HipKernelError(f"{name} failed (status {rc}): {msg.decode() if msg else '?'}")
    return rc


_inited_devices = set()


def init_device(index: int) -> None:
    """edm_init(device) once per device: allocates the library's per-device constants (the zero page) outside any
    launch function, so that every compute entry point is allocation-free and capturable from its first call."""
    if index not in _inited_devices:
        call("edm_init", int(index))
        _inited_devices.add(index)
