"""Tensor-level wrappers over the C ABI: torch only owns the device memory and the stream.

Activations are channels-last-3d tensors of shape [B, T, H, W, C] (contiguous; C may be a padded channel count),
dtype torch.float16 or torch.bfloat16, on a ROCm device.  Nothing here computes with torch ops."""
from dataclasses import dataclass
from typing import Optional, Tuple

import os

import torch

from . import _lib as L

_DT = {torch.float16: L.F16, torch.bfloat16: L.BF16, torch.float32: L.F32}


# fp32 models (the reference's default when from_pretrained gets no torch_dtype) run in SPLIT PRECISION: float tensors in HBM,
# every product as three fp16 MFMAs (include/cvvae.h CVVAE_F32; csrc/conv_kernel.h XP): ~1e-6 relative error, 3x the MFMA work.
SUPPORTS_FP32 = True
# ... or, per model (`fp32_mode = "fast"`), with the two correction terms on the fp8 matrix pipe (CVVAE_F32Q; conv_kernel.h
# XP == 2): ~6e-5 relative error at 2x the MFMA time of a 16-bit model -- the cheapest mode inside north_star's 1e-3 bound
SUPPORTS_FP32_FAST = True
# ... and, for the convolutions behind a GroupNorm + SiLU (whose operand has a bound the host knows), with the correction terms in the
# 6-bit e3m2 format at four times the fp16 rate (CVVAE_F32Q6; XP == 3): 1.5x the MFMA time of a 16-bit model (`fast="fp6"`)
SUPPORTS_FP32_FP6 = True


def _pack_dt(w: torch.Tensor, taps_hw: int, fast) -> int:
    """dtype code a weight is packed with: fp32 weights of multi-tap convolutions take the fast layout when asked to
    (fast = True: bf8 corrections, any operand; fast = "fp6": e3m2 corrections, the launch then needs PackedConv.act_bound)"""
    if fast and w.dtype == torch.float32 and taps_hw > 1:
        return L.F32Q6 if fast == "fp6" else L.F32Q
    return _dt(w.dtype)


def _dt(t: torch.dtype) -> int:
    if t not in _DT:
        raise TypeError(f"the MI355X path runs fp16 / bf16 models on MFMA and fp32 models in split precision; got {t}")
    return _DT[t]


def _xpm(t: torch.dtype) -> int:
    """packed records per (k16, tap): 3 in the split-precision layout of fp32 weights"""
    return 3 if t == torch.float32 else 1


def _wscale(w: torch.Tensor) -> float:
    """fp32 weights are packed as fp16 hi + lo of w * 2^k (k chosen so that max|w| * 2^k is in [512, 1024): the lo parts stay in
    fp16's normal range and sums of up to 27 folded taps cannot overflow); the conv's alpha divides the accumulators by 2^k."""
    if w.dtype != torch.float32:
        return 1.0
    m = float(w.detach().abs().max())
    if not (m > 0.0) or m != m or m == float("inf"):
        return 1.0
    import math
    return float(2.0 ** (9 - math.floor(math.log2(m)) ))


def _stream(t: torch.Tensor) -> int:
    """the HIP stream the launch goes to: the current stream of the TENSOR's device.  A kernel can only be queued on a stream of
    the runtime's current device, so a tensor on another device is refused here (the model-level entry points -- _Net.forward,
    blend_h / blend_v -- switch to the input's device themselves; direct callers wrap the call in torch.cuda.device(t.device))."""
    dev = t.device
    if dev.index is not None and dev.index != torch.cuda.current_device():
        raise RuntimeError(f"cvvae_amd.ops: tensor on {dev} but the current device is cuda:{torch.cuda.current_device()}; "
                           "wrap the call in `with torch.cuda.device(t.device):`")
    return torch.cuda.current_stream(dev).cuda_stream


def _need_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("cvvae_amd ops run on an MI355X (ROCm) device only; there is no CPU path. "
                           f"Got a tensor on {t.device}.")


def kchunk(k: Tuple[int, int, int]) -> int:
    return {(3, 3, 3): 16, (1, 3, 3): 32, (1, 1, 1): 128, (3, 3, 1): 16, (3, 1, 1): 32}[tuple(k)]


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class PackedConv:
    w: torch.Tensor          # packed, fragment order (opaque bytes); [batch, bytes] when batch_stride > 0
    bias: torch.Tensor       # fp32, padded to a multiple of 32
    cout: int
    cin: int                 # padded input channels the kernel consumes (multiple of kchunk)
    k: Tuple[int, int, int]
    cin_real: int = 0        # channels of the source weight (algorithmic FLOP accounting)
    folded: bool = False     # packed by pack_weight_upfold: only valid with conv(..., upsample2x=2)
    batch_stride: int = 0    # bytes between the packed weights of consecutive batch items (pack_weight_batched)
    alg_taps: int = 0        # taps of the REFERENCE op when the packed weights are a folded form (0: kT*kH*kW)
    time_folds: bool = False  # packed with the time-fold slots (pack_weight_tfolds / pack_weight_upfold(time_folds=True))
    wscale: float = 1.0       # fp32 (split-precision) weights were packed as w * wscale (a power of two); conv() undoes it
    dt: int = -1              # dtype code of the packed layout (L.F32 / L.F32Q / L.F32Q6 for fp32 weights); -1: that of the input tensor
    act_bound: float = 0.0    # L.F32Q6 weights: upper bound of |operand| after the prologue (set by whoever knows the GroupNorm affine)


def pack_weight(w: torch.Tensor, bias: Optional[torch.Tensor], k: Tuple[int, int, int], cin_pad: Optional[int] = None,
                strides: Optional[Tuple[int, int, int]] = None, cout: Optional[int] = None, cin: Optional[int] = None,
                out: Optional[torch.Tensor] = None, fold: Tuple[int, int] = (1, 0), offset: int = 0,
                wscale: Optional[float] = None, fast: bool = False) -> PackedConv:
    """Pack a conv / linear weight ([Cout, Cin, *k] contiguous, or any strided view described by `strides` =
    (s_co, s_ci, s_tap) in elements) into MFMA fragment order for cvvae_conv_fwd."""
    lib = L.load()
    _need_gpu(w)
    dt = _pack_dt(w, k[1] * k[2], fast)
    taps = k[0] * k[1] * k[2]
    if strides is None:
        w = w.contiguous()
        cout_, cin_ = w.shape[0], w.shape[1]
        strides = (cin_ * taps, taps, 1)
    else:
        cout_, cin_ = cout, cin
    ck = kchunk(k)
    cin_pad = round_up(cin_, ck) if cin_pad is None else cin_pad
    nbytes = lib.cvvae_packed_weight_bytes(cout_, cin_pad, taps * _xpm(w.dtype))
    if out is None:
        out = torch.zeros(nbytes, dtype=torch.uint8, device=w.device)
    assert out.numel() * out.element_size() >= nbytes
    # wscale (fp32 weights only): force the power-of-two pack scale, e.g. the scale of the conv a fused shortcut accumulates with
    ws = _wscale(w) if (wscale is None or w.dtype != torch.float32) else float(wscale)
    if ws != 1.0:
        w = w * ws
    # fold = (n, stride): every packed element is the sum of n source elements `stride` apart (coinciding taps);
    # offset: element offset of the first source element (e.g. the centre time tap)
    L.check(lib.cvvae_pack_weights_fold(dt, w.data_ptr() + offset * w.element_size(), cout_, cin_, taps, strides[0],
                                        strides[1], strides[2], fold[0], fold[1], cin_pad, ck, out.data_ptr(), _stream(w)),
            "cvvae_pack_weights_fold")
    b = torch.zeros(round_up(cout_, 32), dtype=torch.float32, device=w.device)
    if bias is not None:
        b[:cout_] = bias.detach().to(torch.float32)
    return PackedConv(out, b, cout_, cin_pad, tuple(k), cin_, wscale=ws, dt=dt)


def ncdhw_to_rowpack(x: torch.Tensor, dtype: torch.dtype, pad_mode_w: int) -> torch.Tensor:
    """x: [B,C<=4,T,H,W] -> the row-packed first-layer input [B,T,H,W+3,4] `dtype` (include/cvvae.h, cvvae_conv_desc.in_overlap):
    stored pixel xp holds input pixel xp - 1, columns 0 and W+1 the W padding (replicate / zero), 4 channel slots.  The returned
    tensor is a view of a buffer that stays readable 32 bytes past its end (conv(..., row_packed=True) reads 4 pixels per pixel)."""
    lib = L.load()
    _need_gpu(x)
    x = x.contiguous()
    B, C, T, H, W = x.shape
    if x.dtype not in _DT:
        raise TypeError(f"unsupported input dtype {x.dtype}")
    n = B * T * H * (W + 3) * 4
    buf = torch.empty(n + 16, dtype=dtype, device=x.device)
    L.check(lib.cvvae_ncdhw_to_rowpack(_DT[x.dtype], _dt(dtype), x.data_ptr(), B, C, T, H, W, pad_mode_w, buf.data_ptr(), _stream(x)),
            "cvvae_ncdhw_to_rowpack")
    return buf[:n].view(B, T, H, W + 3, 4)


def ndhwc_to_rowpack(x: torch.Tensor, c: int, pad_mode_w: int) -> torch.Tensor:
    """x: NDHWC [B,T,H,W,Cs] (fp16 / bf16, contiguous), its first c <= 4 channels -> the row-packed first-layer input (see
    ncdhw_to_rowpack): the entry of the device-side pixel pre-processing (modeling.encode_frames_u8) into the same conv_in."""
    lib = L.load()
    _need_gpu(x)
    assert x.dim() == 5 and x.is_contiguous() and x.shape[-1] >= c
    B, T, H, W, Cs = x.shape
    n = B * T * H * (W + 3) * 4
    buf = torch.empty(n + 16, dtype=x.dtype, device=x.device)
    L.check(lib.cvvae_ndhwc_to_rowpack(_dt(x.dtype), x.data_ptr(), B, c, T, H, W, Cs, pad_mode_w, buf.data_ptr(), _stream(x)),
            "cvvae_ndhwc_to_rowpack")
    return buf[:n].view(B, T, H, W + 3, 4)


def pack_weight_rowpack(w: torch.Tensor, bias: Optional[torch.Tensor], time_folds: bool = False) -> PackedConv:
    """[Cout, C<=4, 3, 3, 3] first-layer weight -> packed (3,3,1) weights over the 16 virtual channels (dx, c) of the row-packed
    input: virtual channel dx*4 + c of tap (kt, kh) is w[:, c, kt, kh, dx] (dx = 3 and channel slots >= C are zero)."""
    co, ci = w.shape[0], w.shape[1]
    assert w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and ci <= 4 and w.dtype in (torch.float16, torch.bfloat16)
    wv = torch.zeros((co, 4, 4, 3, 3), dtype=w.dtype, device=w.device)      # [co, dx, c, kt, kh]
    wv[:, :3, :ci] = w.detach().permute(0, 4, 1, 2, 3)                      # [co, dx(kw), c, kt, kh]
    wv = wv.reshape(co, 16, 3, 3, 1).contiguous()
    pw = pack_weight_tfolds(wv, bias, cin_pad=16) if time_folds else pack_weight(wv.reshape(co, 16, 9), bias, (3, 3, 1), cin_pad=16)
    pw.cin_real, pw.alg_taps = ci, 27   # algorithmic FLOP accounting: the 3x3x3 conv over C channels it replaces
    return pw


def pack_weight_tapsn(w: torch.Tensor, time_folds: bool = False) -> PackedConv:
    """[Cout, Cin, 3, 3, 3] last-layer weight (9 * Cout <= 32) -> packed (3,1,1) weights whose output column (dy*3+dx)*Cout + co is
    w[co, :, dt, dy, dx]: the nine spatial taps in the GEMM's N axis (include/cvvae.h cvvae_conv_out_gather).  No bias: the gather
    pass adds it."""
    co, ci = w.shape[0], w.shape[1]
    assert w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and 9 * co <= 32
    wv = torch.zeros((32, ci, 3, 1, 1), dtype=w.dtype, device=w.device)                 # (columns 9*co .. 31: zero weights)
    wv[:9 * co] = w.detach().permute(3, 4, 0, 1, 2).reshape(9 * co, ci, 3, 1, 1)          # [(dy, dx, co), ci, dt, 1, 1]
    pw = pack_weight_tfolds(wv, None) if time_folds else pack_weight(wv.reshape(32, ci, 3), None, (3, 1, 1))
    return pw


def conv_out_gather(v: torch.Tensor, cout: int, bias: torch.Tensor, pad_mode_hw: int, dtype: torch.dtype, u8: bool = False) -> torch.Tensor:
    """v: fp32 [B,T,H,W,ldv] columns of the taps-in-N last layer (conv(..., pack_weight_tapsn, out_f32=True)) -> the layer's output
    [B,cout,T,H,W] `dtype`, or (u8, B = 1) the scripts' uint8 frames [T,H,W,3] of it; bias: fp32 [>= cout]."""
    lib = L.load()
    _need_gpu(v)
    assert v.dim() == 5 and v.is_contiguous() and v.dtype == torch.float32 and bias.dtype == torch.float32 and bias.numel() >= cout
    B, T, H, W, ldv = v.shape
    if u8:
        out = torch.empty((T, H, W, cout), dtype=torch.uint8, device=v.device)
    else:
        out = torch.empty((B, cout, T, H, W), dtype=dtype, device=v.device)
    L.check(lib.cvvae_conv_out_gather(_dt(dtype), v.data_ptr(), B, T, H, W, cout, ldv, bias.data_ptr(), pad_mode_hw,
                                      None if u8 else out.data_ptr(), out.data_ptr() if u8 else None, _stream(v)),
            "cvvae_conv_out_gather")
    return out


@dataclass
class GNPartials:
    """per-tile (n, mean, M2) records of a conv output, written by the conv's epilogue (cvvae_conv_fwd_gn)"""
    buf: torch.Tensor   # fp32, rows * groups * slabs records of 3 floats (layout private to the library: [row][group][slab])
    rows: int
    slabs: int
    C: int
    groups: int
    frames: int = 0     # > 0: written by a per-frame conv (kT = 1) over this many frames: per-frame statistics can be merged from it


def pack_weight_batched(w: torch.Tensor, k: Tuple[int, int, int], cin_pad: int, strides: Tuple[int, int, int], cout: int,
                        cin: int) -> PackedConv:
    """w: [batch, ...] (contiguous per item); item i is packed from w[i] read with `strides` (s_co, s_ci, s_tap): one
    launch for the per-frame K / V^T matrices of the attention blocks.  Used with conv() on a [batch, ...] input."""
    lib = L.load()
    _need_gpu(w)
    dt = _dt(w.dtype)
    assert w.is_contiguous()
    batch = w.shape[0]
    taps = k[0] * k[1] * k[2]
    per = round_up(lib.cvvae_packed_weight_bytes(cout, cin_pad, taps * _xpm(w.dtype)), 16)
    out = torch.zeros((batch, per), dtype=torch.uint8, device=w.device)
    L.check(lib.cvvae_pack_weights_batched(dt, w.data_ptr(), batch, w[0].numel(), cout, cin, taps, strides[0], strides[1],
                                           strides[2], cin_pad, kchunk(k), out.data_ptr(), per, _stream(w)),
            "cvvae_pack_weights_batched")
    b = torch.zeros(round_up(cout, 32), dtype=torch.float32, device=w.device)
    return PackedConv(out, b, cout, cin_pad, tuple(k), cin, batch_stride=per)


def pack_weight_tfolds(w: torch.Tensor, bias: Optional[torch.Tensor], cin_pad: Optional[int] = None, fast: bool = False) -> PackedConv:
    """[Cout, Cin, 3, kH, kW] weight -> packed weights with the three time-fold slots appended (cvvae_pack_weights_tfolds):
    for convs with REPLICATE time padding, where boundary frames read one stored frame through two or three time taps."""
    lib = L.load()
    _need_gpu(w)
    dt = _pack_dt(w, w.shape[3] * w.shape[4], fast)
    assert w.dim() == 5 and w.shape[2] == 3
    w = w.contiguous()
    co, ci, _, kh, kw = w.shape
    nsp = kh * kw
    k = (3, kh, kw)
    ck = kchunk(k)
    cin_pad = round_up(ci, ck) if cin_pad is None else cin_pad
    out = torch.zeros(lib.cvvae_packed_weight_bytes(co, cin_pad, 6 * nsp * _xpm(w.dtype)), dtype=torch.uint8, device=w.device)
    ws = _wscale(w)
    if ws != 1.0:
        w = w * ws
    L.check(lib.cvvae_pack_weights_tfolds(dt, w.data_ptr(), co, ci, nsp, ci * 3 * nsp, 3 * nsp, 1, cin_pad, ck, out.data_ptr(),
                                          _stream(w)), "cvvae_pack_weights_tfolds")
    b = torch.zeros(round_up(co, 32), dtype=torch.float32, device=w.device)
    if bias is not None:
        b[:co] = bias.detach().to(torch.float32)
    return PackedConv(out, b, co, cin_pad, k, ci, time_folds=True, wscale=ws, dt=dt)


def pack_weight_t1(w: torch.Tensor, bias: Optional[torch.Tensor], mode: str, cin_pad: Optional[int] = None,
                   fast: bool = False) -> PackedConv:
    """[Cout, Cin, 3, kH, kW] weight -> the 1 x kH x kW weight a single-frame input sees: mode 'sum' (replicate time padding:
    all three time taps read the one frame) or 'center' (zero time padding: only the centre tap reads data)."""
    assert w.dim() == 5 and w.shape[2] == 3 and mode in ("sum", "center")
    w = w.contiguous()
    co, ci, _, kh, kw = w.shape
    hw = kh * kw
    pw = pack_weight(w, bias, (1, kh, kw), cin_pad=cin_pad, strides=(ci * 3 * hw, 3 * hw, 1), cout=co, cin=ci,
                     fold=(3, hw) if mode == "sum" else (1, 0), offset=0 if mode == "sum" else hw, fast=fast)
    pw.alg_taps = 3 * hw
    return pw


def pack_weight_upfold(w: torch.Tensor, bias: Optional[torch.Tensor], tfold: int = 0, time_folds: bool = False,
                       fast: bool = False) -> PackedConv:
    """Upsample3D's conv weight [Cout, Cin, 3, 3, 3] -> the four folded 3x2x2 phase weights (cvvae_pack_weights_upfold) for
    conv(..., upsample2x=2).  tfold 1 / 2 (single-frame input: time taps summed / centre tap only): 1x2x2 phases."""
    lib = L.load()
    _need_gpu(w)
    dt = _pack_dt(w, 4, fast)
    w = w.contiguous()
    cout_, cin_ = w.shape[0], w.shape[1]
    assert w.numel() == cout_ * cin_ * 27
    cin_pad = round_up(cin_, 32)
    assert not (tfold and time_folds)
    per = lib.cvvae_packed_weight_bytes(cout_, cin_pad, (24 if time_folds else (4 if tfold else 12)) * _xpm(w.dtype))
    out = torch.zeros(4 * per, dtype=torch.uint8, device=w.device)
    ws = _wscale(w)
    if ws != 1.0:
        w = w * ws
    if time_folds:  # 12 taps per phase + the time-fold slots (replicate time padding)
        L.check(lib.cvvae_pack_weights_upfold_tfolds(dt, w.data_ptr(), cout_, cin_, cin_pad, out.data_ptr(), _stream(w)),
                "cvvae_pack_weights_upfold_tfolds")
    else:
        L.check(lib.cvvae_pack_weights_upfold(dt, w.data_ptr(), cout_, cin_, cin_pad, tfold, out.data_ptr(), _stream(w)),
                "cvvae_pack_weights_upfold")
    b = torch.zeros(round_up(cout_, 32), dtype=torch.float32, device=w.device)
    if bias is not None:
        b[:cout_] = bias.detach().to(torch.float32)
    return PackedConv(out, b, cout_, cin_pad, (1, 3, 3) if tfold else (3, 3, 3), cin_, folded=True, alg_taps=27,
                      time_folds=time_folds, wscale=ws, dt=dt)


def conv(x: torch.Tensor, pw: PackedConv, *, stride=(1, 1, 1), pad=((0, 0), (0, 0), (0, 0)), pad_mode_t=L.PAD_ZERO,
         pad_mode_hw=L.PAD_ZERO, prologue=L.PRO_NONE, gn: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
         gn_per_frame=False, residual: Optional[torch.Tensor] = None, upsample2x=False, out_mode=L.OUT_NDHWC,
         shortcut: Optional[Tuple[torch.Tensor, "PackedConv"]] = None, bias: Optional[torch.Tensor] = None,
         out_f32=False, alpha=1.0, out: Optional[torch.Tensor] = None, cout_pad: Optional[int] = None, gn_out: int = 0,
         row_packed: bool = False, act_bound_dev: Optional[torch.Tensor] = None):
    """x: [B,T,H,W,Cs] with Cs >= pw.cin.  pad = ((t_front,t_back),(h_front,h_back),(w_front,w_back)).
    Returns [B,To,Ho,Wo,Cout(_pad)] (NDHWC), [B,2To-1,Ho,Wo,Cout/2] (TIME_SHUFFLE) or [B,Cout,To,Ho,Wo] (NCDHW).
    gn_out = G > 0: also returns the GNPartials of the stored tensor for a following G-group GroupNorm (gn_finalize).
    row_packed: x is ncdhw_to_rowpack()'s [B,T,H,W+3,4] tensor and pw pack_weight_rowpack()'s (3,3,1) weights: the networks' first
    layer with its kW taps folded into 16 virtual channels; pad = (time pad, H pad, (0, 0)); the output has W columns.
    shortcut = (x2, pw2): fused 1x1 shortcut -- out = conv(x) + pw2 . x2 + bias, x2 [B,T,H,W,C2] (same pixels as x), pw2 a
    packed (1,1,1) weight; `bias` then overrides pw.bias (the caller passes b_conv + b_shortcut, fp32, padded to 32)."""
    lib = L.load()
    _need_gpu(x)
    assert x.dim() == 5 and x.is_contiguous()
    dt = _dt(x.dtype)
    if row_packed:
        assert pw.k == (3, 3, 1) and x.shape[-1] == 4 and pw.cin == 16 and pad[2] == (0, 0) and prologue == L.PRO_NONE
        assert residual is None and shortcut is None and not upsample2x and stride[2] == 1
    else:
        assert pw.k != (3, 3, 1), "(3,3,1) weights are the row-packed first layer: conv(..., row_packed=True)"
    if dt == L.F32 and pw.dt in (L.F32Q, L.F32Q6):  # the layout the weights were packed in selects the fp32 arithmetic of this launch
        if shortcut is not None:
            raise ValueError("fast-fp32 weights (CVVAE_F32Q) have no fused-shortcut kernel: run the 1x1 shortcut as its own launch")
        if pw.dt == L.F32Q6 and act_bound_dev is None and not (pw.act_bound > 0.0 and prologue == L.PRO_GN_SILU):
            raise ValueError("fp6-correction weights (CVVAE_F32Q6) go with the GroupNorm + SiLU prologue and a PackedConv.act_bound > 0, "
                             "or with a device-side bound of the operand (act_bound_dev)")
        if act_bound_dev is not None:
            assert pw.dt == L.F32Q6 and act_bound_dev.dtype == torch.float32 and act_bound_dev.numel() == 1 and act_bound_dev.device == x.device
        dt = pw.dt
    B, Ti, Hi, Wi, Cs = x.shape
    assert row_packed or Cs >= pw.cin, f"input has {Cs} channels, packed weights consume {pw.cin}"
    kT, kH, kW = pw.k
    Hl, Wl = (2 * Hi, 2 * Wi) if upsample2x else (Hi, Wi)
    To = (Ti + pad[0][0] + pad[0][1] - kT) // stride[0] + 1
    Ho = (Hl + pad[1][0] + pad[1][1] - kH) // stride[1] + 1
    Wo = (Wl + pad[2][0] + pad[2][1] - kW) // stride[2] + 1
    if row_packed:
        Wo = Wi - 3  # the stored rows carry the W padding (+ one read-ahead pixel)
    cout = pw.cout
    d = L.ConvDesc()
    d.dtype = dt
    d.B, d.Ti, d.Hi, d.Wi, d.Cin = B, Ti, Hi, Wi, pw.cin
    d.in_pix_stride = Cs
    d.in_overlap = 1 if row_packed else 0
    d.act_bound = float(pw.act_bound) if (dt == L.F32Q6 and act_bound_dev is None) else 0.0
    d.act_bound_dev = act_bound_dev.data_ptr() if (dt == L.F32Q6 and act_bound_dev is not None) else None
    if pw.folded != (upsample2x == 2):
        raise ValueError("folded upsample weights (pack_weight_upfold) go with upsample2x=2 and only with it")
    d.upsample2x = int(upsample2x)
    d.kT, d.kH, d.kW = kT, kH, kW
    d.sT, d.sH, d.sW = stride
    d.pad_t, d.pad_h, d.pad_w = pad[0][0], pad[1][0], pad[2][0]
    d.pad_mode_t, d.pad_mode_hw = pad_mode_t, pad_mode_hw
    d.prologue = prologue
    d.gn_rows_per_batch = Ti if gn_per_frame else 1
    d.To, d.Ho, d.Wo, d.Cout = To, Ho, Wo, cout
    d.out_mode = out_mode
    d.out_f32 = 1 if out_f32 else 0
    d.alpha = alpha / pw.wscale
    d.w_batch_stride = pw.batch_stride
    d.w_time_folds = 1 if pw.time_folds else 0
    d.four_wave = 1 if four_wave() else 0
    if shortcut is not None:
        x2, pw2 = shortcut
        assert residual is None and x2.shape[:4] == x.shape[:4] and x2.is_contiguous() and x2.dtype == x.dtype
        assert pw2.k == (1, 1, 1) and pw2.cout == cout and x2.shape[-1] >= pw2.cin
        if pw2.wscale != pw.wscale:
            raise ValueError("fused shortcut on an fp32 model: both weights accumulate in one register set, so they must be packed "
                             "with the same power-of-two scale (pack_weight(..., wscale=pw.wscale))")
        d.sc_Cin, d.sc_in_pix_stride = pw2.cin, x2.shape[-1]
    bias_t = pw.bias if bias is None else bias
    assert bias_t.dtype == torch.float32 and bias_t.numel() >= round_up(cout, 32)
    assert pw.batch_stride == 0 or pw.w.shape[0] == B, "batched weights: one packed set per batch item of the input"
    odt = torch.float32 if out_f32 else x.dtype  # fp32 models: every tensor is float
    if out_mode == L.OUT_NCDHW:
        shape = (B, cout, To, Ho, Wo)
        d.out_pix_stride = 0
    elif out_mode == L.OUT_TIME_SHUFFLE:
        shape = (B, 2 * To - 1, Ho, Wo, cout // 2)
        d.out_pix_stride = cout // 2
    else:
        cp = cout if cout_pad is None else cout_pad
        shape = (B, To, Ho, Wo, cp)
        d.out_pix_stride = cp
    if out is None:
        # padded output channels (cout_pad > cout) must read as zero for the consumer
        out = (torch.zeros if (out_mode == L.OUT_NDHWC and shape[-1] != cout) else torch.empty)(shape, dtype=odt, device=x.device)
    else:
        assert tuple(out.shape) == shape and out.dtype == odt and out.is_contiguous()
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == x.dtype and residual.is_contiguous()
    gsc = gsh = None
    if prologue != L.PRO_NONE:
        gsc, gsh = gn
        assert gsc.dtype == torch.float32 and gsc.shape[-1] == pw.cin and gsc.is_contiguous() and gsh.is_contiguous()
    part = None
    if gn_out:
        slabs = lib.cvvae_conv_gn_slabs(d, gn_out)
        if slabs <= 0:
            L.check(int(slabs), "cvvae_conv_gn_slabs")
        cst = cout // 2 if out_mode == L.OUT_TIME_SHUFFLE else cout
        # time-shuffle outputs: workgroups whose whole tile is the dropped frame exit without writing their records
        alloc = torch.zeros if out_mode == L.OUT_TIME_SHUFFLE else torch.empty
        part = GNPartials(alloc((B, gn_out, slabs, 3), dtype=torch.float32, device=x.device), B, int(slabs), cst, gn_out,
                          frames=To if (kT == 1 and out_mode == L.OUT_NDHWC and not upsample2x) else 0)

    def launch():
        if shortcut is not None:
            L.check(lib.cvvae_conv_fwd_gn_sc(d, x.data_ptr(), pw.w.data_ptr(), bias_t.data_ptr(),
                                             gsc.data_ptr() if gsc is not None else None,
                                             gsh.data_ptr() if gsh is not None else None, shortcut[0].data_ptr(),
                                             shortcut[1].w.data_ptr(), out.data_ptr(), gn_out,
                                             part.buf.data_ptr() if part is not None else None, _stream(x)),
                    "cvvae_conv_fwd_gn_sc")
            return
        L.check(lib.cvvae_conv_fwd_gn(d, x.data_ptr(), pw.w.data_ptr(), bias_t.data_ptr(),
                                      residual.data_ptr() if residual is not None else None,
                                      gsc.data_ptr() if gsc is not None else None, gsh.data_ptr() if gsh is not None else None,
                                      out.data_ptr(), gn_out, part.buf.data_ptr() if part is not None else None, _stream(x)),
                "cvvae_conv_fwd_gn")

    if PROFILE is None:
        launch()
    else:
        PROFILE(d, pw, launch)
    return (out, part) if gn_out else out


def gn_finalize(part: GNPartials, gamma: torch.Tensor, beta: torch.Tensor, eps: float, frames: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
    """merge the conv-epilogue statistics into the (scale, shift) fp32 tables [rows, C] the next conv's prologue consumes.
    frames > 1: per-FRAME tables [rows * frames, C] (row = sample * frames + frame) -- only for records of a per-frame conv (kT = 1)
    over `frames` frames (part.frames says so)."""
    lib = L.load()
    assert gamma.dtype == torch.float32 and gamma.numel() == part.C and beta.numel() == part.C
    assert frames == 1 or (part.frames == frames and part.slabs % frames == 0), "per-frame statistics need a per-frame producer"
    scale = torch.empty((part.rows * frames, part.C), dtype=torch.float32, device=part.buf.device)
    shift = torch.empty((part.rows * frames, part.C), dtype=torch.float32, device=part.buf.device)
    L.check(lib.cvvae_gn_finalize_frames(part.buf.data_ptr(), part.rows, frames, part.slabs, part.C, part.groups, eps, gamma.data_ptr(),
                                         beta.data_ptr(), scale.data_ptr(), shift.data_ptr(), _stream(part.buf)), "cvvae_gn_finalize_frames")
    return scale, shift


# Optional launch observer used by bench.py's roofline pass: called as PROFILE(desc, packed, launch) where launch()
# performs the kernel launch on the current stream.  None (the default) = launch directly.
PROFILE = None


def four_wave() -> bool:
    """are the four-wave conv instances (two workgroups resident per CU; csrc/conv_table.h G11) candidates of a launch's instance
    choice?  A per-launch descriptor field (cvvae_conv_desc.four_wave, ABI 13): the library keeps no state.  On unless
    CVVAE_FOUR_WAVE=0 (read per call: the GPU tests flip it inside one process).  +6 % on the per-frame 128-channel conv at full
    resolution; their fused GroupNorm records are checked bit for bit under co-residency by
    tests/test_gpu_round6.py::test_four_wave_records_are_reproducible_under_co_residency and tools/probes/nw4_stress.py."""
    return os.environ.get("CVVAE_FOUR_WAVE", "1") != "0"


def conv_kernel_name(d: "L.ConvDesc") -> Optional[str]:
    n = L.load().cvvae_conv_kernel_name(d)
    return n.decode() if n else None


def gn_stats(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, groups: int = 32,
             per_frame: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """GroupNorm statistics of x [B,T,H,W,C] -> (scale, shift) fp32 tables [rows, C]; rows = B (5-D GroupNorm,
    stats over T,H,W) or B*T (per-frame).  gamma/beta: fp32 [C]."""
    lib = L.load()
    _need_gpu(x)
    assert x.dim() == 5 and x.is_contiguous()
    B, T, H, W, C = x.shape
    rows, S = (B * T, H * W) if per_frame else (B, T * H * W)
    assert gamma.dtype == torch.float32 and gamma.numel() == C and beta.numel() == C
    scale = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    shift = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.cvvae_gn_workspace_bytes(rows, groups, S), dtype=torch.uint8, device=x.device)
    L.check(lib.cvvae_gn_stats(_dt(x.dtype), x.data_ptr(), rows, S, C, C, groups, eps, gamma.data_ptr(), beta.data_ptr(),
                               scale.data_ptr(), shift.data_ptr(), ws.data_ptr(), _stream(x)), "cvvae_gn_stats")
    return scale, shift


def gn_silu_apply(x: torch.Tensor, gn: Tuple[torch.Tensor, torch.Tensor], silu: bool = True, per_frame: bool = False) -> torch.Tensor:
    """act(x * scale + shift) once per element (cvvae_gn_silu_apply): x [B,T,H,W,C], (scale, shift) fp32 tables [rows, C] with
    rows = B (5-D GroupNorm) or B*T (per frame).  The values equal what conv(..., prologue=PRO_GN_SILU / PRO_GN, gn=gn) stages."""
    lib = L.load()
    _need_gpu(x)
    assert x.dim() == 5 and x.is_contiguous()
    B, T, H, W, C = x.shape
    rows, S = (B * T, H * W) if per_frame else (B, T * H * W)
    sc, sh = gn
    assert sc.dtype == torch.float32 and tuple(sc.shape) == (rows, C) and sc.is_contiguous() and sh.is_contiguous()
    out = torch.empty_like(x)
    L.check(lib.cvvae_gn_silu_apply(_dt(x.dtype), x.data_ptr(), rows, S, C, C, sc.data_ptr(), sh.data_ptr(), 1 if silu else 0,
                                    out.data_ptr(), _stream(x)), "cvvae_gn_silu_apply")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    lib = L.load()
    _need_gpu(x)
    assert x.is_contiguous()
    C = x.shape[-1]
    out = torch.empty_like(x)
    L.check(lib.cvvae_layernorm(_dt(x.dtype), x.data_ptr(), x.numel() // C, C, eps, gamma.data_ptr(), beta.data_ptr(),
                                out.data_ptr(), _stream(x)), "cvvae_layernorm")
    return out


def softmax_rows(s: torch.Tensor, n_valid: int, dtype: torch.dtype, ld_p: Optional[int] = None) -> torch.Tensor:
    """s: [rows, ld_s] fp32 -> [rows, ld_p] dtype; columns >= n_valid are written as 0."""
    lib = L.load()
    _need_gpu(s)
    assert s.dim() == 2 and s.is_contiguous() and s.dtype == torch.float32
    rows, ld_s = s.shape
    ld_p = ld_s if ld_p is None else ld_p
    p = torch.empty((rows, ld_p), dtype=dtype, device=s.device)
    L.check(lib.cvvae_softmax_rows(_dt(dtype), s.data_ptr(), rows, n_valid, ld_s, p.data_ptr(), ld_p, _stream(s)),
            "cvvae_softmax_rows")
    return p


def transpose(x: torch.Tensor, ncols: Optional[int] = None, ld_out: Optional[int] = None) -> torch.Tensor:
    """x: [batch, R, ld] contiguous -> [batch, ncols, ld_out]: the transpose of x[:, :, :ncols] (default: all columns) with rows
    padded to ld_out >= R elements (default R); the padding reads as zero."""
    lib = L.load()
    _need_gpu(x)
    assert x.dim() == 3 and x.is_contiguous()
    b, R, ld = x.shape
    C = ld if ncols is None else ncols
    ldo = R if ld_out is None else ld_out
    assert 0 < C <= ld and ldo >= R
    out = (torch.empty if ldo == R else torch.zeros)((b, C, ldo), dtype=x.dtype, device=x.device)
    L.check(lib.cvvae_transpose(_dt(x.dtype), x.data_ptr(), b, R, C, ld, R * ld, out.data_ptr(), ldo, C * ldo, _stream(x)),
            "cvvae_transpose")
    return out


def gn_bwd_input(x: torch.Tensor, gy: torch.Tensor, tabs: Tuple[torch.Tensor, torch.Tensor], gamma: torch.Tensor,
                 beta: torch.Tensor, silu: bool, add: Optional[torch.Tensor] = None, per_frame: bool = False,
                 groups: int = 32) -> torch.Tensor:
    """Gradient w.r.t. x of act(GroupNorm(x)) given gy = dL/d(act(...)) (cvvae_gn_bwd_input): x, gy, add [B,T,H,W,C];
    tabs = (rstd, -mean*rstd) fp32 tables [rows, C] = gn_finalize / gn_stats with gamma 1, beta 0; gamma / beta fp32 [C].
    `add` is summed into the result (the skip branch of a residual block)."""
    lib = L.load()
    _need_gpu(x)
    assert x.dim() == 5 and x.is_contiguous() and gy.is_contiguous() and gy.shape == x.shape and gy.dtype == x.dtype
    B, T, H, W, C = x.shape
    rows, S = (B * T, H * W) if per_frame else (B, T * H * W)
    rs, nm = tabs
    assert rs.dtype == torch.float32 and tuple(rs.shape) == (rows, C) and tuple(nm.shape) == (rows, C)
    assert rs.is_contiguous() and nm.is_contiguous() and gamma.dtype == torch.float32 and gamma.numel() == C and beta.numel() == C
    if add is not None:
        assert add.shape == x.shape and add.dtype == x.dtype and add.is_contiguous()
    out = torch.empty_like(x)
    ws = torch.empty(max(int(lib.cvvae_gn_bwd_workspace_bytes(rows, groups, S)), 16), dtype=torch.uint8, device=x.device)
    L.check(lib.cvvae_gn_bwd_input(_dt(x.dtype), x.data_ptr(), gy.data_ptr(), add.data_ptr() if add is not None else None, rows, S, C,
                                   groups, rs.data_ptr(), nm.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1 if silu else 0,
                                   out.data_ptr(), ws.data_ptr(), _stream(x)), "cvvae_gn_bwd_input")
    return out


def gn_bwd_input_params(x: torch.Tensor, gy: torch.Tensor, tabs: Tuple[torch.Tensor, torch.Tensor], gamma: torch.Tensor,
                        beta: torch.Tensor, silu: bool, add: Optional[torch.Tensor] = None, per_frame: bool = False,
                        groups: int = 32) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """gn_bwd_input for a trainable norm: (dL/dx, d gamma, d beta) in one reduction + one apply pass (cvvae_gn_bwd_input_params)"""
    lib = L.load()
    _need_gpu(x)
    assert x.dim() == 5 and x.is_contiguous() and gy.is_contiguous() and gy.shape == x.shape and gy.dtype == x.dtype
    B, T, H, W, C = x.shape
    rows, S = (B * T, H * W) if per_frame else (B, T * H * W)
    rs, nm = tabs
    assert rs.dtype == torch.float32 and tuple(rs.shape) == (rows, C) and rs.is_contiguous() and nm.is_contiguous()
    if add is not None:
        assert add.shape == x.shape and add.dtype == x.dtype and add.is_contiguous()
    if C % groups or (C // groups) % 4 or C % 8 or C > 2048 or 256 % (C // 8):
        # (both GroupNorm backward kernels map a 256-thread block onto whole rows of 8-channel lanes; the model classes admit only
        #  widths of 128 * 2^k -- modeling._channel_constraints -- so this is a caller error, reported before the launch)
        raise ValueError(f"GroupNorm backward: C = {C} with {groups} groups is not a width the kernels take (C/8 must divide 256, "
                         f"channels per group a multiple of 4, C <= 2048)")
    out = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = torch.empty(max(int(lib.cvvae_gn_bwd_params_workspace_bytes(rows, groups, S, C)), 16), dtype=torch.uint8, device=x.device)
    L.check(lib.cvvae_gn_bwd_input_params(_dt(x.dtype), x.data_ptr(), gy.data_ptr(), add.data_ptr() if add is not None else None, rows,
                                          S, C, groups, rs.data_ptr(), nm.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                          1 if silu else 0, out.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), _stream(x)),
            "cvvae_gn_bwd_input_params")
    return out, dg, db


def conv_wgrad(a: torch.Tensor, gy: torch.Tensor, k: Tuple[int, int, int], *, stride=(1, 1, 1),
               pad=((0, 0), (0, 0), (0, 0)), pad_mode_t=L.PAD_ZERO, pad_mode_hw=L.PAD_ZERO, cin: Optional[int] = None,
               cout: Optional[int] = None, bias: bool = False):
    """Weight gradient of conv(a, W, stride, pad, modes) given gy = dL/d(output) (cvvae_conv_wgrad).  a: [B,Ti,Hi,Wi,Cs] -- the
    operand the forward multiplied (AFTER its GroupNorm + SiLU, before padding); gy: [B,To,Ho,Wo,Cg].  cin / cout: the weight's real
    channel counts (default: the tensors' last dims).  Returns fp32 [cout, cin, kT, kH, kW] (nn.Conv3d's layout); with bias = True
    the pair (dW, fp32 [cout] bias gradient = sum of gy over the pixels) -- from the same launch where the kernel fuses it
    (cvvae_conv_wgrad_bias), else from bias_grad."""
    lib = L.load()
    _need_gpu(a)
    assert a.dim() == 5 and gy.dim() == 5 and a.is_contiguous() and gy.is_contiguous() and a.dtype == gy.dtype
    B, Ti, Hi, Wi, Cs = a.shape
    _, To, Ho, Wo, Cg = gy.shape
    kT, kH, kW = k
    assert gy.shape[0] == B
    assert To == (Ti + pad[0][0] + pad[0][1] - kT) // stride[0] + 1 and Ho == (Hi + pad[1][0] + pad[1][1] - kH) // stride[1] + 1 and \
        Wo == (Wi + pad[2][0] + pad[2][1] - kW) // stride[2] + 1, "gy does not have the forward conv's output shape"
    cin = Cs if cin is None else cin
    cout = Cg if cout is None else cout
    d = L.ConvDesc()
    d.dtype = _dt(a.dtype)
    d.B, d.Ti, d.Hi, d.Wi = B, Ti, Hi, Wi
    d.Cin = round_up(cin, 8)
    assert d.Cin <= Cs
    d.in_pix_stride = Cs
    d.kT, d.kH, d.kW = kT, kH, kW
    d.sT, d.sH, d.sW = stride
    d.pad_t, d.pad_h, d.pad_w = pad[0][0], pad[1][0], pad[2][0]
    d.pad_mode_t, d.pad_mode_hw = pad_mode_t, pad_mode_hw
    d.To, d.Ho, d.Wo = To, Ho, Wo
    d.Cout = round_up(cout, 8)
    assert d.Cout <= Cg
    nb = int(lib.cvvae_conv_wgrad_workspace_bytes(d))
    if nb <= 0:
        L.check(nb, "cvvae_conv_wgrad_workspace_bytes")
    ws = torch.empty(nb, dtype=torch.uint8, device=a.device)
    dw = torch.empty((d.Cout, d.Cin, kT * kH * kW), dtype=torch.float32, device=a.device)
    db = None
    if bias and lib.cvvae_conv_wgrad_fuses_bias(d) == 1:
        db = torch.empty(d.Cout, dtype=torch.float32, device=a.device)
    L.check(lib.cvvae_conv_wgrad_bias(d, a.data_ptr(), gy.data_ptr(), Cg, dw.data_ptr(), db.data_ptr() if db is not None else None,
                                      ws.data_ptr(), _stream(a)), "cvvae_conv_wgrad_bias")
    dw = dw[:cout, :cin].reshape(cout, cin, kT, kH, kW)
    if not bias:
        return dw
    return dw, (db[:cout] if db is not None else bias_grad(gy, cout=cout))


def bias_grad(gy: torch.Tensor, cout: Optional[int] = None) -> torch.Tensor:
    """sum of gy [..., C] over every pixel -> fp32 [cout] (a conv's bias gradient; cvvae_channel_sums with x = NULL)"""
    lib = L.load()
    _need_gpu(gy)
    assert gy.is_contiguous()
    C = gy.shape[-1]
    S = gy.numel() // C
    cc = round_up(C if cout is None else cout, 8)
    assert cc <= C
    ws = torch.empty(max(int(lib.cvvae_channel_sums_workspace_bytes(1, S, cc)), 16), dtype=torch.uint8, device=gy.device)
    out = torch.empty(cc, dtype=torch.float32, device=gy.device)
    L.check(lib.cvvae_channel_sums(_dt(gy.dtype), None, gy.data_ptr(), C, 1, S, cc, None, None, None, None, 0, out.data_ptr(), None,
                                   ws.data_ptr(), _stream(gy)), "cvvae_channel_sums")
    return out[:C if cout is None else cout]


def gn_bwd_params(x: torch.Tensor, gy: torch.Tensor, tabs: Tuple[torch.Tensor, torch.Tensor], gamma: torch.Tensor,
                  beta: torch.Tensor, silu: bool, per_frame: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """(d gamma, d beta) fp32 [C] of act(GroupNorm(x)) given gy = dL/d(act(...)): the affine half of
    aten::native_group_norm_backward (cvvae_channel_sums); arguments as gn_bwd_input."""
    lib = L.load()
    _need_gpu(x)
    assert x.dim() == 5 and x.is_contiguous() and gy.is_contiguous() and gy.shape == x.shape and gy.dtype == x.dtype
    B, T, H, W, C = x.shape
    rows, S = (B * T, H * W) if per_frame else (B, T * H * W)
    rs, nm = tabs
    assert rs.dtype == torch.float32 and tuple(rs.shape) == (rows, C) and rs.is_contiguous() and nm.is_contiguous()
    ws = torch.empty(max(int(lib.cvvae_channel_sums_workspace_bytes(rows, S, C)), 16), dtype=torch.uint8, device=x.device)
    db = torch.empty(C, dtype=torch.float32, device=x.device)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    L.check(lib.cvvae_channel_sums(_dt(x.dtype), x.data_ptr(), gy.data_ptr(), C, rows, S, C, rs.data_ptr(), nm.data_ptr(),
                                   gamma.data_ptr(), beta.data_ptr(), 1 if silu else 0, db.data_ptr(), dg.data_ptr(), ws.data_ptr(),
                                   _stream(x)), "cvvae_channel_sums")
    return dg, db


def pad_fold(gp: torch.Tensor, pad_t: Tuple[int, int], pad_hw: int, pad_mode_t: int, pad_mode_hw: int,
             add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gp: gradient w.r.t. the PADDED input [B, T+pt0+pt1, H+2p, W+2p, C] -> gradient w.r.t. the input [B,T,H,W,C]: the adjoint of
    the forward's out-of-range coordinate map (replicate: borders collect their pad region; zero: the interior) (+ add)."""
    lib = L.load()
    _need_gpu(gp)
    assert gp.dim() == 5 and gp.is_contiguous()
    B, Tp, Hp, Wp, C = gp.shape
    T, H, W = Tp - pad_t[0] - pad_t[1], Hp - 2 * pad_hw, Wp - 2 * pad_hw
    out = torch.empty((B, T, H, W, C), dtype=gp.dtype, device=gp.device)
    if add is not None:
        assert add.shape == out.shape and add.dtype == gp.dtype and add.is_contiguous()
    L.check(lib.cvvae_pad_fold(_dt(gp.dtype), gp.data_ptr(), B, T, H, W, C, pad_t[0], pad_t[1], pad_hw, pad_hw, pad_mode_t, pad_mode_hw,
                               add.data_ptr() if add is not None else None, out.data_ptr(), _stream(gp)), "cvvae_pad_fold")
    return out


def softmax_bwd_rows(p: torch.Tensor, gp: torch.Tensor, n_valid: int, alpha: float, ld_o: Optional[int] = None) -> torch.Tensor:
    """p: [rows, ld_p] probabilities (softmax_rows), gp: [rows, ld_g] fp32 -> alpha * p * (gp - rowsum(p * gp)) as [rows, ld_o] of
    p's dtype, columns >= n_valid written as 0."""
    lib = L.load()
    _need_gpu(p)
    assert p.dim() == 2 and gp.dim() == 2 and p.is_contiguous() and gp.is_contiguous() and gp.dtype == torch.float32
    assert p.shape[0] == gp.shape[0]
    rows, ld_p = p.shape
    ld_o = ld_p if ld_o is None else ld_o
    out = torch.empty((rows, ld_o), dtype=p.dtype, device=p.device)
    L.check(lib.cvvae_softmax_bwd_rows(_dt(p.dtype), p.data_ptr(), ld_p, gp.data_ptr(), gp.shape[1], rows, n_valid, float(alpha),
                                       out.data_ptr(), ld_o, _stream(p)), "cvvae_softmax_bwd_rows")
    return out


def upsample2x_sum(g: torch.Tensor) -> torch.Tensor:
    """g: [N,1,2H,2W,C] -> [N,1,H,W,C]: the gradient of the nearest-neighbour x2 upsample (sum over every 2x2 block)."""
    lib = L.load()
    _need_gpu(g)
    assert g.dim() == 5 and g.shape[1] == 1 and g.is_contiguous() and g.shape[2] % 2 == 0 and g.shape[3] % 2 == 0
    N, _, H2, W2, C = g.shape
    out = torch.empty((N, 1, H2 // 2, W2 // 2, C), dtype=g.dtype, device=g.device)
    L.check(lib.cvvae_upsample2x_sum(_dt(g.dtype), g.data_ptr(), N, H2 // 2, W2 // 2, C, out.data_ptr(), _stream(g)),
            "cvvae_upsample2x_sum")
    return out


def attention_d512(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, n: int, scale: float) -> torch.Tensor:
    """softmax(q k^T * scale) v per batch item in ONE launch (cvvae_attention_d512): q, k [batch, N, 512] fp16 / bf16 (contiguous),
    vt = transpose(v, ld_out=round_up(N, 32)) [batch, 512, ldvt] with zero padding -> [batch, N, 512]."""
    lib = L.load()
    _need_gpu(q)
    assert q.dim() == 3 and q.shape[-1] == 512 and q.shape == k.shape and q.is_contiguous() and k.is_contiguous() and vt.is_contiguous()
    assert vt.shape[0] == q.shape[0] and vt.shape[1] == 512 and vt.shape[2] >= round_up(n, 32) and q.shape[1] == n
    out = torch.empty_like(q)
    L.check(lib.cvvae_attention_d512(_dt(q.dtype), q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), q.shape[0], n, vt.shape[2],
                                     float(scale), _stream(q)), "cvvae_attention_d512")
    return out


def temporal_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """q,k,v: NDHWC [B,T,H,W,C] -> softmax(q k^T / sqrt(C)) v over the T frames of every pixel (T <= 8)."""
    lib = L.load()
    _need_gpu(q)
    assert q.dim() == 5 and q.is_contiguous() and k.is_contiguous() and v.is_contiguous() and q.shape == k.shape == v.shape
    B, T, H, W, C = q.shape
    out = torch.empty_like(q)
    L.check(lib.cvvae_temporal_attention(_dt(q.dtype), q.data_ptr(), k.data_ptr(), v.data_ptr(), B, T, H * W, C,
                                         out.data_ptr(), _stream(q)), "cvvae_temporal_attention")
    return out


def temporal_attention_bwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, go: torch.Tensor):
    """gradients (gq, gk, gv) of temporal_attention(q, k, v) given go = dL/d(out); all NDHWC [B,T,H,W,C], T <= 8"""
    lib = L.load()
    _need_gpu(q)
    assert q.dim() == 5 and all(t.is_contiguous() and t.shape == q.shape and t.dtype == q.dtype for t in (q, k, v, go))
    B, T, H, W, C = q.shape
    gq, gk, gv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    L.check(lib.cvvae_temporal_attention_bwd(_dt(q.dtype), q.data_ptr(), k.data_ptr(), v.data_ptr(), go.data_ptr(), B, T, H * W, C,
                                             gq.data_ptr(), gk.data_ptr(), gv.data_ptr(), _stream(q)), "cvvae_temporal_attention_bwd")
    return gq, gk, gv


def layernorm_bwd(x: torch.Tensor, gy: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float):
    """backward of layernorm(x) over the last axis: (dL/dx, d gamma, d beta).  A LayerNorm over C is a one-group GroupNorm whose
    rows are the tokens, so this is gn_stats + gn_bwd_input + gn_bwd_params on the [tokens,1,1,1,C] view.  The kernels take
    their rows on grid y (<= 65535): more tokens (the vae3d decoder's temporal attention at 512x512 crops is 20480 per sample, so a
    batch of 4 is beyond it) run as several launches over row chunks, the affine sums added in chunk order."""
    C = x.shape[-1]
    n = x.numel() // C
    assert x.is_contiguous() and gy.is_contiguous() and gy.shape == x.shape, (tuple(x.shape), tuple(gy.shape))
    one = torch.ones(C, dtype=torch.float32, device=x.device)
    zero = torch.zeros(C, dtype=torch.float32, device=x.device)
    x2, g2 = x.view(n, C), gy.view(n, C)
    gx = torch.empty_like(x2)
    dg = db = None
    for a in range(0, n, LAYERNORM_ROWS):
        m = min(LAYERNORM_ROWS, n - a)
        x5, g5 = x2[a:a + m].view(m, 1, 1, 1, C), g2[a:a + m].view(m, 1, 1, 1, C)
        tabs = gn_stats(x5, one, zero, eps, groups=1)
        gx[a:a + m] = gn_bwd_input(x5, g5, tabs, gamma, beta, silu=False, groups=1).view(m, C)
        dgc, dbc = gn_bwd_params(x5, g5, tabs, gamma, beta, silu=False)
        dg, db = (dgc, dbc) if dg is None else (dg + dgc, db + dbc)
    return gx.view(x.shape), dg, db


LAYERNORM_ROWS = 65535  # rows one launch of the GroupNorm kernels takes (grid y)


def ncdhw_to_ndhwc(x: torch.Tensor, cpad: int, dtype: torch.dtype) -> torch.Tensor:
    """x: [B,C,T,H,W] (fp16/bf16/fp32) -> [B,T,H,W,cpad] `dtype`, pad channels zero."""
    lib = L.load()
    _need_gpu(x)
    x = x.contiguous()
    B, C, T, H, W = x.shape
    if x.dtype not in _DT:
        raise TypeError(f"unsupported input dtype {x.dtype}")
    out = torch.empty((B, T, H, W, cpad), dtype=dtype, device=x.device)
    L.check(lib.cvvae_ncdhw_to_ndhwc(_DT[x.dtype], _dt(dtype), x.data_ptr(), B, C, T, H, W, cpad, out.data_ptr(), _stream(x)),
            "cvvae_ncdhw_to_ndhwc")
    return out


def ndhwc_to_ncdhw(x: torch.Tensor, c: int) -> torch.Tensor:
    lib = L.load()
    _need_gpu(x)
    assert x.dim() == 5 and x.is_contiguous()
    B, T, H, W, Cs = x.shape
    out = torch.empty((B, c, T, H, W), dtype=x.dtype, device=x.device)
    L.check(lib.cvvae_ndhwc_to_ncdhw(_dt(x.dtype), x.data_ptr(), B, c, T, H, W, Cs, out.data_ptr(), _stream(x)),
            "cvvae_ndhwc_to_ncdhw")
    return out


def blend_(a: torch.Tensor, b: torch.Tensor, overlap: int, axis: int) -> torch.Tensor:
    """in place on b (NCDHW tensors): axis 0 = blend_v (H), 1 = blend_h (W).  modeling_vae.py:321-341."""
    lib = L.load()
    _need_gpu(b)
    assert a.dim() == 5 and b.dim() == 5 and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype
    rows = b.shape[0] * b.shape[1] * b.shape[2]
    assert a.shape[:3] == b.shape[:3]
    L.check(lib.cvvae_blend(_dt(b.dtype), a.data_ptr(), a.shape[3], a.shape[4], b.data_ptr(), b.shape[3], b.shape[4], rows,
                            overlap, axis, _stream(b)), "cvvae_blend")
    return b


def resize_tables(in_size: int, out_size: int):
    """Tables of ONE axis of torch's antialiased bilinear interpolation of uint8 tensors (what `transforms.Resize` runs on the scripts'
    uint8 clip, cvvae_inference_video.py:14-16,28): per output position the first input position, the tap count and the int16-scaled
    triangle-filter weights, and the weights' precision.  Plain Python (host logic; CPU-tested bit for bit against
    F.interpolate(uint8, mode="bilinear", antialias=True), tests/test_host_logic.py) -> (xmin, xsize, weights [out, ksize], ksize, precision)."""
    import math
    scale = in_size / out_size
    support = scale if scale >= 1.0 else 1.0
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    xmin, xsize, wf = [], [], []
    for i in range(out_size):
        center = scale * (i + 0.5)
        lo = max(0, int(center - support + 0.5))
        hi = min(in_size, int(center + support + 0.5))
        ws = [max(0.0, 1.0 - abs((j + lo - center + 0.5) * invscale)) for j in range(hi - lo)]
        tot = sum(ws)
        xmin.append(lo)
        xsize.append(hi - lo)
        wf.append([v / tot for v in ws] + [0.0] * (ksize - (hi - lo)))
    mx = max(max(r) for r in wf)
    prec = 0
    while prec < 22 and int(0.5 + mx * (1 << (prec + 1))) < (1 << 15):
        prec += 1
    wi = [[int(0.5 + v * (1 << prec)) for v in r] for r in wf]  # (triangle weights are never negative)
    return xmin, xsize, wi, ksize, prec


def resize_frames_u8(frames: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """uint8 frames [T,H,W,C] -> [T,size[0],size[1],C]: the scripts' `transforms.Resize(size=(height, width))` (antialiased bilinear on
    uint8, width pass then height pass, fixed-point weights) on the device; bit-exact against torch's CPU kernel."""
    lib = L.load()
    _need_gpu(frames)
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.is_contiguous()
    T, H, W, C = frames.shape
    oh, ow = int(size[0]), int(size[1])
    cur = frames
    for axis, (n_in, n_out) in ((2, (W, ow)), (1, (H, oh))):
        if n_in == n_out:
            continue
        xmin, xsize, wi, ksize, prec = resize_tables(n_in, n_out)
        dev = frames.device
        t_xmin = torch.tensor(xmin, dtype=torch.int32, device=dev)
        t_xsize = torch.tensor(xsize, dtype=torch.int32, device=dev)
        t_w = torch.tensor(wi, dtype=torch.int32, device=dev).contiguous()
        shp = list(cur.shape)
        outer = 1
        for d in shp[:axis]:
            outer *= d
        inner = 1
        for d in shp[axis + 1:]:
            inner *= d
        shp[axis] = n_out
        out = torch.empty(shp, dtype=torch.uint8, device=dev)
        L.check(lib.cvvae_resize_u8_axis(cur.data_ptr(), out.data_ptr(), outer, n_in, n_out, inner, t_xmin.data_ptr(), t_xsize.data_ptr(),
                                         t_w.data_ptr(), ksize, prec, _stream(frames)), "cvvae_resize_u8_axis")
        cur = out
    return cur


def frames_u8_to_ndhwc(frames: torch.Tensor, cpad: int, dtype: torch.dtype) -> torch.Tensor:
    """uint8 frames [T,H,W,3] -> [1,T,H,W,cpad] dtype = u8/127.5 - 1 (the scripts' normalisation, in dtype arithmetic)."""
    lib = L.load()
    _need_gpu(frames)
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[-1] == 3 and frames.is_contiguous()
    T, H, W, _ = frames.shape
    out = torch.empty((1, T, H, W, cpad), dtype=dtype, device=frames.device)
    L.check(lib.cvvae_frames_u8_to_ndhwc(_dt(dtype), frames.data_ptr(), T * H * W, cpad, out.data_ptr(), _stream(frames)),
            "cvvae_frames_u8_to_ndhwc")
    return out


def ncdhw_to_frames_u8(x: torch.Tensor) -> torch.Tensor:
    """decoder output [1,3,T,H,W] -> uint8 frames [T,H,W,3] = u8((clamp(x,-1,1)+1)*127.5) (the scripts' post-processing)."""
    lib = L.load()
    _need_gpu(x)
    assert x.dim() == 5 and x.shape[0] == 1 and x.shape[1] == 3 and x.is_contiguous()
    _, _, T, H, W = x.shape
    out = torch.empty((T, H, W, 3), dtype=torch.uint8, device=x.device)
    L.check(lib.cvvae_ncdhw_to_frames_u8(_dt(x.dtype), x.data_ptr(), T * H * W, out.data_ptr(), _stream(x)),
            "cvvae_ncdhw_to_frames_u8")
    return out
