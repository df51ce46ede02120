"""The C-ABI library must load WITHOUT a GPU and export every function include/cvvae.h declares, with the
argument checks reachable (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "cvvae.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(cvvae_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if n != "cvvae_conv_kchunk"))  # static inline helper


def test_library_exports_every_declared_symbol():
    from cvvae_amd import _lib
    lib = _lib.load()
    names = declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/cvvae.h but not exported"
        assert n in _lib.PROTOTYPES, f"{n} has no ctypes prototype in cvvae_amd/_lib.py"
    assert lib.cvvae_abi_version() == _lib.ABI_VERSION


def test_struct_layout_matches_header():
    from cvvae_amd import _lib
    # natural alignment of the C struct: 6 x i32, i64, 20 x i32, i64, f32 (+pad), i64, 2 x i32, i64, i32 + f32, i32 (+pad), pointer = 176 bytes
    # (ABI 13: `four_wave` and `act_bound_dev` appended; everything before it where it was since ABI 9)
    assert ctypes.sizeof(_lib.ConvDesc) == 176 and _lib.ConvDesc.w_batch_stride.offset == 128
    assert _lib.ConvDesc.four_wave.offset == 160 and _lib.ConvDesc.act_bound_dev.offset == 168
    assert _lib.ConvDesc.in_overlap.offset == 152 and _lib.ConvDesc.act_bound.offset == 156
    assert _lib.ConvDesc.sc_Cin.offset == 136 and _lib.ConvDesc.sc_in_pix_stride.offset == 144
    assert _lib.ConvDesc.in_pix_stride.offset == 24 and _lib.ConvDesc.out_pix_stride.offset == 112
    assert _lib.ConvDesc.alpha.offset == 120


def test_argument_checks_without_gpu():
    from cvvae_amd import _lib
    lib = _lib.load()
    assert lib.cvvae_conv_fwd(None, None, None, None, None, None, None, None, None) == -1
    d = _lib.ConvDesc()
    assert lib.cvvae_conv_kernel_name(d) is None
    d.dtype, d.B, d.Ti, d.Hi, d.Wi, d.Cin, d.in_pix_stride = 1, 1, 5, 16, 16, 128, 128
    d.kT = d.kH = d.kW = 3
    d.sT = d.sH = d.sW = 1
    d.To, d.Ho, d.Wo, d.Cout, d.out_pix_stride, d.gn_rows_per_batch = 5, 16, 16, 128, 128, 1
    assert lib.cvvae_conv_kernel_name(d).decode().startswith("conv_k333_s111")
    # fp6-correction instances: fp32 tensors, GroupNorm + SiLU prologue, and a bound of the operand from the caller
    d.dtype, d.prologue, d.pad_t, d.pad_h, d.pad_w, d.pad_mode_t, d.pad_mode_hw = _lib.F32Q6, 1, 1, 1, 1, 1, 1
    assert lib.cvvae_conv_kernel_name(d) is None  # act_bound missing
    d.act_bound = 8.0
    assert lib.cvvae_conv_kernel_name(d).decode().endswith("_xq6")
    d.prologue = 0
    assert lib.cvvae_conv_kernel_name(d) is None  # no instance without the prologue
    d.dtype, d.prologue = _lib.F32Q, 0
    assert lib.cvvae_conv_kernel_name(d) is None  # a bound with any other dtype is an argument error
    d.act_bound, d.dtype = 0.0, 1
    d.kT = 5
    assert lib.cvvae_conv_fwd(d, 1, 1, 1, None, None, None, 1, None) == -2  # unsupported kernel size
    assert lib.cvvae_packed_weight_bytes(128, 128, 27) == 128 * 128 * 27 * 2 + 16384
    assert lib.cvvae_gn_stats(1, None, 1, 1, 128, 128, 32, 1e-6, None, None, None, None, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    from cvvae_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcvvae_hip.so")
    with pytest.raises(_lib.CvvaeError, match="no CPU/eager fallback"):
        _lib.load()


def _wgrad_desc(_lib, dtype, cin, cout, k, stride, T, H, W):
    d = _lib.ConvDesc()
    d.dtype, d.B, d.Ti, d.Hi, d.Wi, d.Cin, d.in_pix_stride = dtype, 1, T, H, W, cin, cin
    d.kT, d.kH, d.kW = k
    d.sT, d.sH, d.sW = stride
    d.pad_t, d.pad_h, d.pad_w = k[0] - 1, k[1] // 2, k[2] // 2
    d.To, d.Ho, d.Wo, d.Cout = (T - 1) // stride[0] + 1, (H - 1) // stride[1] + 1, (W - 1) // stride[2] + 1, cout
    return d


def test_weight_gradient_plan_without_gpu():
    """cvvae_conv_wgrad_workspace_bytes is pure host code: it pins the plan of the weight-gradient launch -- the number of slabs the
    pixel range is cut into (round 5's rounds rule for the DMA kernel: the fewest rounds of 32 workgroups per XCD that whole slabs
    fill to >= 90 %; 768 workgroups for the register-staged kernel) -- and which launches also produce the bias gradient (ABI 12)."""
    from cvvae_amd import _lib
    lib = _lib.load()

    def slabs(d):
        n_co, n_ci, taps = (d.Cout + 127) // 128, (d.Cin + 63) // 64, d.kT * d.kH * d.kW
        tile = taps * n_co * 128 * n_ci * 64 * 4
        nb = int(lib.cvvae_conv_wgrad_workspace_bytes(d))
        assert nb > 0
        # partial tiles (a multiple of 256 bytes) + the 512-byte zero page + the per-slab bias sums (nslab x Coutp floats, 256-aligned)
        for n in range(1, 4097):
            if (n * tile + 255) // 256 * 256 + 512 + (n * n_co * 128 * 4 + 255) // 256 * 256 == nb:
                return n
        raise AssertionError(f"workspace of {nb} bytes is not a whole number of slabs")

    BF16, F32 = _lib.BF16, _lib.F32
    # 128 -> 128 3x3x3: 6 members per slab -> 5 slabs per XCD in ONE round of 32 workgroups per XCD
    assert slabs(_wgrad_desc(_lib, BF16, 128, 128, (3, 3, 3), (1, 1, 1), 17, 256, 256)) == 40
    # per-frame 3x3: 2 members -> 16 slabs per XCD in one round
    assert slabs(_wgrad_desc(_lib, BF16, 128, 128, (1, 3, 3), (1, 1, 1), 17, 256, 256)) == 128
    # 256 -> 256 (24 members) and 512 -> 512 (96): three rounds = the 768 workgroups of the old rule
    assert slabs(_wgrad_desc(_lib, BF16, 256, 256, (3, 3, 3), (1, 1, 1), 9, 128, 128)) == 32
    assert slabs(_wgrad_desc(_lib, BF16, 512, 512, (3, 3, 3), (1, 1, 1), 9, 64, 64)) == 8
    # 128 -> 256 (12 members): two rounds, 5 slabs per XCD
    assert slabs(_wgrad_desc(_lib, BF16, 128, 256, (3, 3, 3), (1, 1, 1), 9, 128, 128)) == 40
    # fp32 models and 1x1 layers stay on the register-staged kernel's 768-workgroup rule
    assert slabs(_wgrad_desc(_lib, F32, 128, 128, (3, 3, 3), (1, 1, 1), 17, 256, 256)) == 128
    assert slabs(_wgrad_desc(_lib, BF16, 512, 512, (1, 1, 1), (1, 1, 1), 1, 32, 32)) == 24  # (32 members: 768 / 32)
    # the bias gradient rides along exactly where the DMA kernel runs
    assert lib.cvvae_conv_wgrad_fuses_bias(_wgrad_desc(_lib, BF16, 128, 128, (3, 3, 3), (1, 1, 1), 5, 64, 64)) == 1
    assert lib.cvvae_conv_wgrad_fuses_bias(_wgrad_desc(_lib, _lib.F16, 128, 128, (1, 3, 3), (1, 2, 2), 5, 64, 64)) == 1
    assert lib.cvvae_conv_wgrad_fuses_bias(_wgrad_desc(_lib, F32, 128, 128, (3, 3, 3), (1, 1, 1), 5, 64, 64)) == 0
    assert lib.cvvae_conv_wgrad_fuses_bias(_wgrad_desc(_lib, BF16, 512, 512, (1, 1, 1), (1, 1, 1), 1, 32, 32)) == 0
    assert lib.cvvae_conv_wgrad_fuses_bias(None) < 0
