"""Test infrastructure: a plain-PyTorch (CPU, fp32) stand-in for the subset of `cvvae_amd.ops` that the 2-D constraint decoder
and its input-gradient pass call, so that the HOST logic of engine.constraint_decoder2d / grad.constraint_decoder2d_backward
(what is taped, which weights are transposed, which tensor feeds which launch) is checked against torch.autograd without a GPU.
It emulates each op's documented arithmetic -- it is not a fallback: nothing in the product imports it (tests only)."""
import contextlib
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F

from cvvae_amd import _lib as L
from cvvae_amd import ops


@dataclass
class FakePacked:
    w: torch.Tensor            # [cout, cin_real, taps] fp32, or [batch, cout, cin_real] (batched 1x1)
    bias: torch.Tensor         # fp32 [round_up(cout, 32)]
    cout: int
    cin: int                   # padded K
    k: tuple
    cin_real: int
    folded: bool = False
    batch_stride: int = 0
    time_folds: bool = False
    wscale: float = 1.0
    alg_taps: int = 0


@dataclass
class FakePart:
    x: torch.Tensor            # the stored tensor the statistics describe, [B,T,H,W,C]
    rows: int
    C: int
    groups: int


def _bias(cout, bias, dev):
    b = torch.zeros(ops.round_up(cout, 32), dtype=torch.float32, device=dev)
    if bias is not None:
        b[:cout] = bias.detach().float()
    return b


def pack_weight(w, bias, k, cin_pad=None, strides=None, cout=None, cin=None, fold=(1, 0), offset=0, out=None, wscale=None):
    assert strides is None and fold == (1, 0) and offset == 0
    taps = k[0] * k[1] * k[2]
    co, ci = w.shape[0], w.shape[1]
    ck = ops.kchunk(k)
    cp = ops.round_up(ci, ck) if cin_pad is None else cin_pad
    return FakePacked(w.detach().float().reshape(co, ci, taps), _bias(co, bias, w.device), co, cp, tuple(k), ci)


def pack_weight_upfold(w, bias, tfold=0, time_folds=False):
    assert tfold == 2 and not time_folds  # Upsample2D: the centre time tap of an otherwise zero 3x3x3 weight
    co, ci = w.shape[0], w.shape[1]
    return FakePacked(w.detach().float()[:, :, 1].reshape(co, ci, 9), _bias(co, bias, w.device), co, ops.round_up(ci, 32), (1, 3, 3),
                      ci, folded=True)


def pack_weight_batched(w, k, cin_pad, strides, cout, cin):
    assert tuple(k) == (1, 1, 1) and w.is_contiguous()
    b = w.shape[0]
    flat = w.detach().float().reshape(b, -1)
    idx = (torch.arange(cout)[:, None] * strides[0] + torch.arange(cin)[None, :] * strides[1]).reshape(-1)
    m = flat[:, idx].reshape(b, cout, cin)
    return FakePacked(m, _bias(cout, None, w.device), cout, cin_pad, (1, 1, 1), cin, batch_stride=1)


def _stats(x, rows, groups, eps):
    """x [rows, S, C] fp32 -> mean, rstd [rows, groups]"""
    r, S, C = x.shape
    g = x.reshape(r, S, groups, C // groups)
    mean = g.mean((1, 3))
    var = g.var((1, 3), unbiased=False)
    return mean, (var + eps).rsqrt()


def _tables(x5, gamma, beta, eps, groups, per_frame):
    B, T, H, W, C = x5.shape
    rows, S = (B * T, H * W) if per_frame else (B, T * H * W)
    mean, rstd = _stats(x5.float().reshape(rows, S, C), rows, groups, eps)
    cpg = C // groups
    scale = gamma[None, :] * rstd.repeat_interleave(cpg, 1)
    shift = beta[None, :] - mean.repeat_interleave(cpg, 1) * scale
    return scale.contiguous(), shift.contiguous()


def gn_stats(x, gamma, beta, eps, groups=32, per_frame=False):
    return _tables(x, gamma, beta, eps, groups, per_frame)


def gn_finalize(part, gamma, beta, eps):
    return _tables(part.x, gamma, beta, eps, part.groups, False)


def conv(x, pw, *, stride=(1, 1, 1), pad=((0, 0), (0, 0), (0, 0)), pad_mode_t=L.PAD_ZERO, pad_mode_hw=L.PAD_ZERO,
         prologue=L.PRO_NONE, gn=None, gn_per_frame=False, residual=None, upsample2x=False, out_mode=L.OUT_NDHWC, shortcut=None,
         bias=None, out_f32=False, alpha=1.0, out=None, cout_pad=None, gn_out=0):
    assert stride == (1, 1, 1) and pad_mode_hw == L.PAD_ZERO and pw.folded == (upsample2x == 2)
    B, T, H, W, Cs = x.shape
    assert Cs >= pw.cin, (Cs, pw.cin)
    a = x.float()[..., :pw.cin_real]
    if prologue != L.PRO_NONE:
        sc, sh = gn
        rows = B * T if gn_per_frame else B
        assert tuple(sc.shape) == (rows, pw.cin) and pw.cin == pw.cin_real
        shape = (B, T, 1, 1, pw.cin) if gn_per_frame else (B, 1, 1, 1, pw.cin)
        a = a * sc.reshape(shape) + sh.reshape(shape)
        if prologue == L.PRO_GN_SILU:
            a = F.silu(a)
        a = a.to(x.dtype).float()  # the kernel stages the activation in the storage dtype
    if pw.k == (1, 3, 3):
        assert pad == ((0, 0), (1, 1), (1, 1)) and pw.batch_stride == 0
        f = a.reshape(B * T, H, W, -1).permute(0, 3, 1, 2)
        if upsample2x:
            f = F.interpolate(f, scale_factor=2.0, mode="nearest")
        y = F.conv2d(f, pw.w.reshape(pw.cout, pw.cin_real, 3, 3), None, padding=1).permute(0, 2, 3, 1)
        y = y.reshape(B, T, y.shape[1], y.shape[2], pw.cout)
    else:
        assert pw.k == (1, 1, 1) and pad == ((0, 0), (0, 0), (0, 0))
        if pw.batch_stride:
            assert pw.w.shape[0] == B
            y = torch.einsum("bthwc,boc->bthwo", a, pw.w)
        else:
            y = a @ pw.w[:, :, 0].t()
    y = y * alpha + (pw.bias if bias is None else bias)[:pw.cout]
    if shortcut is not None:
        x2, pw2 = shortcut
        y = y + x2.float()[..., :pw2.cin_real] @ pw2.w[:, :, 0].t()
    if residual is not None:
        y = y + residual.float()
    odt = torch.float32 if out_f32 else x.dtype
    if out_mode == L.OUT_NCDHW:
        res = y.permute(0, 4, 1, 2, 3).contiguous().to(odt)
    else:
        assert out_mode == L.OUT_NDHWC
        cp = pw.cout if cout_pad is None else cout_pad
        res = torch.zeros(*y.shape[:-1], cp, dtype=odt)
        res[..., :pw.cout] = y.to(odt)
    if gn_out:
        return res, FakePart(res, B, pw.cout, gn_out)
    return res


def softmax_rows(s, n_valid, dtype, ld_p=None):
    rows, ld_s = s.shape
    p = torch.zeros(rows, ld_s if ld_p is None else ld_p, dtype=dtype)
    p[:, :n_valid] = torch.softmax(s[:, :n_valid].float(), -1).to(dtype)
    return p


def transpose(x, ncols=None, ld_out=None):
    b, R, ld = x.shape
    C = ld if ncols is None else ncols
    out = torch.zeros(b, C, R if ld_out is None else ld_out, dtype=x.dtype)
    out[:, :, :R] = x[:, :, :C].transpose(1, 2)
    return out


def ncdhw_to_ndhwc(x, cpad, dtype):
    B, C, T, H, W = x.shape
    out = torch.zeros(B, T, H, W, cpad, dtype=dtype)
    out[..., :C] = x.permute(0, 2, 3, 4, 1).to(dtype)
    return out


def gn_bwd_input(x, gy, tabs, gamma, beta, silu, add=None, per_frame=False, groups=32):
    B, T, H, W, C = x.shape
    rows, S = (B * T, H * W) if per_frame else (B, T * H * W)
    rs, nm = tabs
    xf = x.float().reshape(rows, S, C)
    xh = xf * rs[:, None, :] + nm[:, None, :]
    a = xh * gamma + beta
    d = torch.sigmoid(a) * (1 + a * (1 - torch.sigmoid(a))) if silu else torch.ones_like(a)
    gh = gy.float().reshape(rows, S, C) * d * gamma
    cpg = C // groups
    c1 = gh.reshape(rows, S, groups, cpg).mean((1, 3)).repeat_interleave(cpg, 1)[:, None, :]
    c2 = (gh * xh).reshape(rows, S, groups, cpg).mean((1, 3)).repeat_interleave(cpg, 1)[:, None, :]
    gx = (rs[:, None, :] * (gh - c1 - xh * c2)).reshape(x.shape)
    if add is not None:
        gx = gx + add.float()
    return gx.to(x.dtype)


def softmax_bwd_rows(p, gp, n_valid, alpha, ld_o=None):
    rows, ld_p = p.shape
    out = torch.zeros(rows, ld_p if ld_o is None else ld_o, dtype=p.dtype)
    pf, g = p.float()[:, :n_valid], gp[:, :n_valid]
    out[:, :n_valid] = (alpha * pf * (g - (pf * g).sum(-1, keepdim=True))).to(p.dtype)
    return out


def upsample2x_sum(g):
    N, _, H2, W2, C = g.shape
    return g.float().reshape(N, 1, H2 // 2, 2, W2 // 2, 2, C).sum((3, 5)).to(g.dtype)


_NAMES = ["pack_weight", "pack_weight_upfold", "pack_weight_batched", "gn_stats", "gn_finalize", "conv", "softmax_rows",
          "transpose", "ncdhw_to_ndhwc", "gn_bwd_input", "softmax_bwd_rows", "upsample2x_sum"]


@contextlib.contextmanager
def patched():
    saved = {n: getattr(ops, n) for n in _NAMES}
    try:
        for n in _NAMES:
            setattr(ops, n, globals()[n])
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
