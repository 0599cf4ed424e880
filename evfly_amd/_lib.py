"""ctypes binding of libevfly_hip.so (the C ABI declared in include/evfly_hip.h).

There is no CPU fallback: every compute entry point of evfly_amd goes through this
library, and `lib()` raises if it is missing or a GPU is not present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EVFLY_LIB: developer override (A/B of two builds of the same ABI from tools/); the default is the in-tree build
LIB_PATH = os.environ.get("EVFLY_LIB") or os.path.join(_HERE, "libevfly_hip.so")
_LIB = None

c_p = C.c_void_p
c_i = C.c_int
c_i64 = C.c_int64
c_f = C.c_float
c_d = C.c_double


class ModelConfig(C.Structure):
    """Mirror of `evfly_model_config` (include/evfly_hip.h)."""
    _fields_ = [
        ("has_unet", c_i), ("num_in_channels", c_i), ("num_out_channels", c_i), ("form_bev", c_i),
        ("skip_type", c_i), ("num_recurrent_unet", c_i), ("input_h", c_i), ("input_w", c_i),
        ("evs_min_cutoff", c_f), ("head", c_i),
        ("vit_in_channels", c_i),
        ("vit_width", c_i * 2), ("vit_heads", c_i * 2), ("vit_layers", c_i * 2),
        ("vit_reduction", c_i * 2), ("vit_patch", c_i * 2), ("vit_stride", c_i * 2),
        ("vit_pad", c_i * 2), ("vit_expansion", c_i),
        ("compute_dtype", c_i),
        ("velpred", c_i), ("enc_num_layers", c_i),
        ("enc_kernel", c_i * 4), ("enc_stride", c_i * 4), ("enc_out_channels", c_i * 4), ("enc_act", c_i * 4),
        ("enc_pool_type", c_i), ("enc_pool_kernel", c_i * 4), ("enc_pool_stride", c_i * 4),
        ("enc_invert_pool_inputs", c_i),
        ("fc_num_layers", c_i), ("fc_size", c_i * 8), ("fc_act", c_i * 8),
        ("is_deployment", c_i), ("velpred_lstm_layers", c_i),
    ]


ACT_CODES = {"none": 0, "relu": 1, "leaky_relu": 2, "tanh": 3, "sigmoid": 4}
POOL_CODES = {"none": 0, "max": 1, "avg": 2}


# name -> (restype, argtypes); must list every symbol include/evfly_hip.h declares
SIGNATURES = {
    "evfly_abi_version": (c_i, []),
    "evfly_last_error": (C.c_char_p, []),
    "evfly_voxelize_windows": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_d, c_d,
                                     c_p, c_p, c_p, c_p]),
    "evfly_voxelize_windows_roi": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_d, c_d,
                                         c_p, c_p, c_p, c_p]),
    "evfly_voxel_prepare": (c_i, [c_p, c_i64, c_p, c_i, c_p, c_i, c_p, c_p, c_p]),
    "evfly_voxelize_windows_prepared": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_d, c_d,
                                              c_p, c_p, c_i, c_p, c_p, c_p, c_p]),
    "evfly_eventframe_rows_f64": (c_i, [c_p, c_i64, c_i, c_i, c_i, c_d, c_d, c_i64, c_d, c_d,
                                        c_p, c_p, c_p, c_p]),
    "evfly_accumulate_u8": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p, c_p]),
    "evfly_accumulate_reset": (c_i, [c_p, c_i64, c_p]),
    "evfly_condition_frames": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p]),
    "evfly_remap_cubic": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "evfly_difflog_events": (c_i, [c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_p, c_p]),
    "evfly_resize_bilinear": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "evfly_model_create": (c_i, [C.POINTER(ModelConfig), C.POINTER(c_p)]),
    "evfly_model_load_tensor": (c_i, [c_p, C.c_char_p, c_p, C.POINTER(c_i64), c_i]),
    "evfly_model_finalize": (c_i, [c_p]),
    "evfly_model_destroy": (None, [c_p]),
    "evfly_model_arena_generation": (c_i, [c_p]),
    "evfly_unet_forward": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "evfly_vit_forward": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    "evfly_vit_stage_forward": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_p]),
    "evfly_vit_block_forward": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_p, c_p]),
    "evfly_e2v_forward": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "evfly_model_tap": (c_i64, [c_p, C.c_char_p, c_p, c_i64, C.POINTER(c_i64), c_p]),
    "evfly_model_set_profiling": (c_i, [c_p, c_i]),
    "evfly_model_set_profile_filter": (c_i, [c_p, C.c_char_p]),
    "evfly_model_profile_count": (c_i, [c_p]),
    "evfly_model_profile_get": (c_i, [c_p, c_i, C.c_char_p, c_i, C.POINTER(c_d), C.POINTER(c_d),
                                      C.POINTER(c_d), C.POINTER(c_i)]),
    "evfly_model_profile_exec_flops": (c_i, [c_p, c_i, C.POINTER(c_d)]),
    "evfly_model_profile_useful_flops": (c_i, [c_p, c_i, C.POINTER(c_d)]),
    "evfly_model_profile_reset": (c_i, [c_p]),
    "evfly_op_conv2d_nhwc": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i,
                                   c_p, c_p, c_i, c_p]),
    "evfly_op_pool2d_nhwc": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "evfly_op_velpred_vec": (c_i, [c_p, c_i64, c_i, c_p, c_p]),
    "evfly_op_grouped_conv_gelu": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p]),
    "evfly_op_mixffn_block_bf16": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "evfly_op_convlstm_gates": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p]),
    "evfly_convlstm_workspace_bytes": (c_i64, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "evfly_convlstm_forward": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i64, c_p]),
    "evfly_convlstm_standby_runs": (c_i64, []),
    "evfly_op_conv2d_nhwc_bf16": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i,
                                        c_p, c_p, c_p]),
}


def load_library():
    """dlopen the library and attach signatures (no GPU needed: used by the symbol test)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"evfly_amd: {LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # torch first: its wheel carries its own libamdhip64 / libhsa-runtime64. If this library (linked
        # against /opt/rocm) were loaded before torch, the process would end up with two HIP runtimes in the
        # wrong binding order and the first HIP call here fails with "no ROCm-capable device is detected".
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def lib():
    """The library, for compute: additionally requires a visible GPU."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("evfly_amd: no MI355X/ROCm device visible; the HIP path is the only path "
                           "(there is no CPU fallback).")
    return load_library()


def check(rc):
    if rc < 0:
        msg = load_library().evfly_last_error().decode(errors="replace")
        raise RuntimeError(f"evfly_hip error {rc}: {msg}")
    return rc


def ptr(t):
    """Device (or host) pointer of a torch tensor / None."""
    if t is None:
        return None
    assert t.is_contiguous(), "evfly_amd: tensor handed to the C ABI must be contiguous"
    return t.data_ptr()


def cur_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
