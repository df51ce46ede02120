"""ctypes binding of libcvvae_hip.so (include/cvvae.h).  There is NO fallback: if the HIP library is missing
the import of any compute entry fails loudly -- build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C cvvae_amd/csrc -j8`."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CVVAE_LIB: tuning aid -- load another build of the same ABI (A/B of kernel variants); the product path is the in-tree file
LIB_PATH = os.environ.get("CVVAE_LIB") or os.path.join(_HERE, "libcvvae_hip.so")

F16, BF16, F32, F32Q, F32Q6 = 0, 1, 2, 3, 4
PAD_ZERO, PAD_REPLICATE = 0, 1
PRO_NONE, PRO_GN_SILU, PRO_GN = 0, 1, 2
OUT_NDHWC, OUT_NCDHW, OUT_TIME_SHUFFLE = 0, 1, 2
ABI_VERSION = 13


class ConvDesc(ctypes.Structure):
    """mirror of `cvvae_conv_desc` (include/cvvae.h) -- field order and types must match."""

    _fields_ = [
        ("dtype", ctypes.c_int32),
        ("B", ctypes.c_int32), ("Ti", ctypes.c_int32), ("Hi", ctypes.c_int32), ("Wi", ctypes.c_int32),
        ("Cin", ctypes.c_int32),
        ("in_pix_stride", ctypes.c_int64),
        ("upsample2x", ctypes.c_int32),
        ("kT", ctypes.c_int32), ("kH", ctypes.c_int32), ("kW", ctypes.c_int32),
        ("sT", ctypes.c_int32), ("sH", ctypes.c_int32), ("sW", ctypes.c_int32),
        ("pad_t", ctypes.c_int32), ("pad_h", ctypes.c_int32), ("pad_w", ctypes.c_int32),
        ("pad_mode_t", ctypes.c_int32), ("pad_mode_hw", ctypes.c_int32),
        ("prologue", ctypes.c_int32),
        ("gn_rows_per_batch", ctypes.c_int32),
        ("To", ctypes.c_int32), ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32), ("Cout", ctypes.c_int32),
        ("out_mode", ctypes.c_int32),
        ("out_f32", ctypes.c_int32),
        ("out_pix_stride", ctypes.c_int64),
        ("alpha", ctypes.c_float),
        ("w_batch_stride", ctypes.c_int64),
        ("sc_Cin", ctypes.c_int32), ("w_time_folds", ctypes.c_int32),
        ("sc_in_pix_stride", ctypes.c_int64),
        ("in_overlap", ctypes.c_int32), ("act_bound", ctypes.c_float),
        ("four_wave", ctypes.c_int32),
        ("act_bound_dev", ctypes.c_void_p),
    ]


_vp, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float

# name -> (restype, argtypes): every symbol include/cvvae.h declares
PROTOTYPES = {
    "cvvae_abi_version": (_i32, []),
    "cvvae_packed_weight_bytes": (ctypes.c_size_t, [_i32, _i32, _i32]),
    "cvvae_pack_weights": (_i32, [_i32, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _vp, _vp]),
    "cvvae_pack_weights_upfold": (_i32, [_i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "cvvae_pack_weights_tfolds": (_i32, [_i32, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _vp, _vp]),
    "cvvae_pack_weights_upfold_tfolds": (_i32, [_i32, _vp, _i32, _i32, _i32, _vp, _vp]),
    "cvvae_pack_weights_batched": (_i32, [_i32, _vp, _i32, _i64, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp]),
    "cvvae_pack_weights_fold": (_i32, [_i32, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _i64, _i32, _i32, _vp, _vp]),
    "cvvae_conv_fwd": (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cvvae_conv_kernel_name": (ctypes.c_char_p, [ctypes.POINTER(ConvDesc)]),
    "cvvae_conv_gn_slabs": (_i64, [ctypes.POINTER(ConvDesc), _i32]),
    "cvvae_conv_fwd_gn": (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "cvvae_conv_fwd_gn_sc": (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "cvvae_gn_finalize": (_i32, [_vp, _i32, _i64, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "cvvae_gn_finalize_frames": (_i32, [_vp, _i32, _i32, _i64, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "cvvae_gn_workspace_bytes": (ctypes.c_size_t, [_i32, _i32, _i64]),
    "cvvae_gn_stats": (_i32, [_i32, _vp, _i32, _i64, _i32, _i64, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cvvae_gn_silu_apply": (_i32, [_i32, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _i32, _vp, _vp]),
    "cvvae_gn_bwd_workspace_bytes": (_i64, [_i32, _i32, _i64]),
    "cvvae_gn_bwd_input": (_i32, [_i32, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "cvvae_gn_bwd_params_workspace_bytes": (_i64, [_i32, _i32, _i64, _i32]),
    "cvvae_gn_bwd_input_params": (_i32, [_i32, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "cvvae_softmax_bwd_rows": (_i32, [_i32, _vp, _i64, _vp, _i64, _i64, _i32, _f32, _vp, _i64, _vp]),
    "cvvae_upsample2x_sum": (_i32, [_i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "cvvae_conv_wgrad_workspace_bytes": (_i64, [ctypes.POINTER(ConvDesc)]),
    "cvvae_conv_wgrad": (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _i64, _vp, _vp, _vp]),
    "cvvae_conv_wgrad_fuses_bias": (_i32, [ctypes.POINTER(ConvDesc)]),
    "cvvae_conv_wgrad_bias": (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "cvvae_channel_sums_workspace_bytes": (_i64, [_i32, _i64, _i32]),
    "cvvae_channel_sums": (_i32, [_i32, _vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "cvvae_temporal_attention_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp, _vp, _vp, _vp]),
    "cvvae_pad_fold": (_i32, [_i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "cvvae_layernorm": (_i32, [_i32, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    "cvvae_softmax_rows": (_i32, [_i32, _vp, _i64, _i32, _i64, _vp, _i64, _vp]),
    "cvvae_transpose": (_i32, [_i32, _vp, _i32, _i32, _i32, _i64, _i64, _vp, _i64, _i64, _vp]),
    "cvvae_temporal_attention": (_i32, [_i32, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp, _vp]),
    "cvvae_ncdhw_to_ndhwc": (_i32, [_i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "cvvae_ncdhw_to_rowpack": (_i32, [_i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "cvvae_conv_out_gather": (_i32, [_i32, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _vp, _i32, _vp, _vp, _vp]),
    "cvvae_attention_d512": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _f32, _vp]),
    "cvvae_resize_u8_axis": (_i32, [_vp, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _i32, _i32, _vp]),
    "cvvae_ndhwc_to_rowpack": (_i32, [_i32, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _vp, _vp]),
    "cvvae_ndhwc_to_ncdhw": (_i32, [_i32, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _vp, _vp]),
    "cvvae_frames_u8_to_ndhwc": (_i32, [_i32, _vp, _i64, _i32, _vp, _vp]),
    "cvvae_ncdhw_to_frames_u8": (_i32, [_i32, _vp, _i64, _vp, _vp]),
    "cvvae_blend": (_i32, [_i32, _vp, _i32, _i32, _vp, _i32, _i32, _i64, _i32, _i32, _vp]),
}

_lib = None


# translation units whose kernels only run in training (weight gradients, channel sums, their reductions): no launch of an
# encode / decode step comes from them, so the counters of the inference bench do not go stale when they change
TRAINING_ONLY_SOURCES = ("wgrad_kernel.hip",)


def source_fingerprint() -> str:
    """sha256 (first 16 hex digits) over the kernel sources the inference path is built from (csrc/*.h, csrc/*.hip except
    TRAINING_ONLY_SOURCES, include/cvvae.h): measurements kept under profiles/ are stamped with it, and bench.py flags them stale
    when the sources have moved on"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.h")) + glob.glob(os.path.join(_HERE, "csrc", "*.hip")))
    files = [f for f in files if os.path.basename(f) not in TRAINING_ONLY_SOURCES]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "cvvae.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


class CvvaeError(RuntimeError):
    pass


def load():
    """dlopen libcvvae_hip.so and type every entry point.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise CvvaeError(
            f"{LIB_PATH} not found: the MI355X HIP extension is not built and there is no CPU/eager fallback. "
            "Run `make -C cvvae_amd/csrc -j8` (or __graft_entry__.build()).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.cvvae_abi_version() != ABI_VERSION:
        raise CvvaeError("libcvvae_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc == -1:
        raise ValueError(f"{what}: invalid argument (CVVAE_EINVAL)")
    if rc == -2:
        raise NotImplementedError(f"{what}: unsupported shape/option (CVVAE_EUNSUPPORTED)")
    raise CvvaeError(f"{what}: HIP error {rc}")
