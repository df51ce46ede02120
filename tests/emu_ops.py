"""Test infrastructure: a plain-PyTorch (CPU, fp32) stand-in for `cvvae_amd.ops`, so that the HOST logic above the C ABI -- the
launch programs of engine.py (which tensor, weight form, padding, prologue, epilogue and output mode every launch gets), the
window / tile wrappers of modeling.py and the input-gradient pass of grad.py -- is checked without a GPU against the reference's
golden vectors and torch.autograd.  Each function restates the documented arithmetic of one op (include/cvvae.h); the folded
weight forms (time folds, single-frame folds, folded upsample) are emulated by the unfolded convolution they equal.
It is not a fallback: nothing in the product imports it (tests only)."""
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
    dt: int = -1
    act_bound: float = 0.0


@dataclass
class FakePart:
    x: torch.Tensor            # the stored tensor the statistics describe, [B,T,H,W,C]
    rows: int
    C: int
    groups: int
    frames: int = 0
    slabs: int = 0


Q6_CALLS = []  # (max |operand|, bound) of every emulated CVVAE_F32Q6 launch


def _bias(cout, bias, dev):
    b = torch.zeros(ops.round_up(cout, 32), dtype=torch.float32, device=dev)
    if bias is not None:
        b[:cout] = bias.detach().float()
    return b


def pack_weight(w, bias, k, cin_pad=None, strides=None, cout=None, cin=None, fold=(1, 0), offset=0, out=None, wscale=None, fast=False):
    assert strides is None and fold == (1, 0) and offset == 0
    taps = k[0] * k[1] * k[2]
    co, ci = w.shape[0], w.shape[1]
    ck = ops.kchunk(k)
    cp = ops.round_up(ci, ck) if cin_pad is None else cin_pad
    # (the power-of-two pack scale of fp32 weights is recorded as the real packer records it -- the host logic reads it, e.g. the
    #  fused-shortcut range check -- but the emulated weights stay unscaled: conv() below never divides by it)
    ws = (ops._wscale(w) if wscale is None else float(wscale)) if w.dtype == torch.float32 else 1.0
    return FakePacked(w.detach().float().reshape(co, ci, taps), _bias(co, bias, w.device), co, cp, tuple(k), ci, wscale=ws,
                      dt=ops._pack_dt(w, k[1] * k[2], fast))


def ncdhw_to_rowpack(x, dtype, pad_mode_w):
    """[B,C,T,H,W] -> [B,T,H,W+3,4]: stored pixel xp = input pixel xp - 1, columns 0 / W+1 = the W padding, column W+2 and channel
    slots >= C zero (include/cvvae.h, cvvae_ncdhw_to_rowpack)"""
    B, C, T, H, W = x.shape
    f = x.to(dtype)
    f = F.pad(f.float(), (1, 1, 0, 0, 0, 0), mode="replicate").to(dtype) if pad_mode_w == L.PAD_REPLICATE else F.pad(f, (1, 1))
    out = torch.zeros(B, T, H, W + 3, 4, dtype=dtype)
    out[:, :, :, :W + 2, :C] = f.permute(0, 2, 3, 4, 1)
    return out


def pack_weight_tapsn(w, time_folds=False):
    co, ci = w.shape[0], w.shape[1]
    wv = torch.zeros((32, ci, 3, 1, 1), dtype=w.dtype)
    wv[:9 * co] = w.detach().permute(3, 4, 0, 1, 2).reshape(9 * co, ci, 3, 1, 1)
    pw = pack_weight(wv.reshape(32, ci, 3), None, (3, 1, 1))
    pw.time_folds = time_folds
    return pw


def conv_out_gather(v, cout, bias, pad_mode_hw, dtype, u8=False):
    """out[co] = bias[co] + sum over (dy, dx) of V[y+dy-1, x+dx-1][(dy*3+dx)*cout + co] with the layer's spatial padding"""
    B, T, H, W, _ = v.shape
    f = v.permute(0, 4, 1, 2, 3)                                                    # [B,32,T,H,W]
    f = F.pad(f, (1, 1, 1, 1, 0, 0), mode="replicate") if pad_mode_hw == L.PAD_REPLICATE else F.pad(f, (1, 1, 1, 1, 0, 0))
    out = bias[:cout].float().view(1, cout, 1, 1, 1).expand(B, cout, T, H, W).clone()
    for dy in range(3):
        for dx in range(3):
            out = out + f[:, (dy * 3 + dx) * cout:(dy * 3 + dx + 1) * cout, :, dy:dy + H, dx:dx + W]
    out = out.to(dtype)
    return ncdhw_to_frames_u8(out) if u8 else out


def ndhwc_to_rowpack(x, c, pad_mode_w):
    return ncdhw_to_rowpack(x[..., :c].permute(0, 4, 1, 2, 3), x.dtype, pad_mode_w)


def pack_weight_rowpack(w, bias, time_folds=False):
    """the (3,3,1) weights over the 16 virtual channels compute the original 3x3x3 conv: keep the original weight"""
    co, ci = w.shape[0], w.shape[1]
    pw = FakePacked(w.detach().float().reshape(co, ci, 27), _bias(co, bias, w.device), co, 16, (3, 3, 1), ci, time_folds=time_folds,
                    alg_taps=27)
    return pw


def pack_weight_tfolds(w, bias, cin_pad=None, fast=False):
    """the summed time slots only change HOW boundary frames are multiplied, not the result: the plain weight"""
    co, ci, _, kh, kw = w.shape
    pw = pack_weight(w.reshape(co, ci, 3 * kh * kw), bias, (3, kh, kw), cin_pad=cin_pad, fast=fast)
    pw.time_folds = True
    return pw


def pack_weight_t1(w, bias, mode, cin_pad=None, fast=False):
    """a single-frame input under replicate ('sum': all three time taps read the frame) / zero ('center') time padding"""
    co, ci, _, kh, kw = w.shape
    w2 = w.detach().float().sum(2) if mode == "sum" else w.detach().float()[:, :, 1]
    pw = pack_weight(w2.reshape(co, ci, kh * kw), bias, (1, kh, kw), cin_pad=cin_pad)
    pw.alg_taps = 3 * kh * kw
    return pw


def pack_weight_upfold(w, bias, tfold=0, time_folds=False, fast=False):
    """nearest-2x + conv with the phase-folded weights == the conv of the upsampled tensor with the plain weight"""
    co, ci = w.shape[0], w.shape[1]
    wf = w.detach().float().reshape(co, ci, 3, 3, 3)
    if tfold == 0:
        raw, k = wf.reshape(co, ci, 27), (3, 3, 3)
    else:
        raw, k = (wf.sum(2) if tfold == 1 else wf[:, :, 1]).reshape(co, ci, 9), (1, 3, 3)
    return FakePacked(raw, _bias(co, bias, w.device), co, ops.round_up(ci, 32), k, ci, folded=True, time_folds=time_folds,
                      alg_taps=27)


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


def gn_finalize(part, gamma, beta, eps, frames=1):
    if frames > 1:
        assert part.frames == frames
    return _tables(part.x, gamma, beta, eps, part.groups, frames > 1)


def _pad3(f, pad, mode_t, mode_hw):
    """f [B,C,T,H,W]: pad H and W (zero / replicate), then T (zero / replicate) -- the order of the reference's F.pad calls"""
    (tf, tb), (hf, hb), (wf, wb) = pad
    if hf or hb or wf or wb:
        f = F.pad(f, (wf, wb, hf, hb, 0, 0), mode="replicate") if mode_hw == L.PAD_REPLICATE else F.pad(f, (wf, wb, hf, hb))
    if tf or tb:
        f = F.pad(f, (0, 0, 0, 0, tf, tb), mode="replicate") if mode_t == L.PAD_REPLICATE else F.pad(f, (0, 0, 0, 0, tf, tb))
    return f


def conv(x, pw, *, stride=(1, 1, 1), pad=((0, 0), (0, 0), (0, 0)), pad_mode_t=L.PAD_ZERO, pad_mode_hw=L.PAD_ZERO,
         prologue=L.PRO_NONE, gn=None, gn_per_frame=False, residual=None, upsample2x=False, out_mode=L.OUT_NDHWC, shortcut=None,
         bias=None, out_f32=False, alpha=1.0, out=None, cout_pad=None, gn_out=0, row_packed=False, act_bound_dev=None):
    assert pw.folded == (upsample2x == 2)
    assert act_bound_dev is None  # (the fp6 form of a fast-fp32 model: device only)
    B, T, H, W, Cs = x.shape
    if row_packed:  # x already carries the W padding: columns 0 .. W+1 of the stored row are the padded input row
        assert pw.k == (3, 3, 1) and Cs == 4 and pad[2] == (0, 0) and prologue == L.PRO_NONE and residual is None and shortcut is None
        f = x.float()[:, :, :, :W - 1, :pw.cin_real].permute(0, 4, 1, 2, 3)
        f = _pad3(f, (pad[0], pad[1], (0, 0)), pad_mode_t, pad_mode_hw)
        y = F.conv3d(f, pw.w.reshape(pw.cout, pw.cin_real, 3, 3, 3), None, stride=stride).permute(0, 2, 3, 4, 1)
        y = y * alpha + (pw.bias if bias is None else bias)[:pw.cout]
        res = y.to(x.dtype)
        return (res, FakePart(res, B, pw.cout, gn_out)) if gn_out else res
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
    if pw.dt == L.F32Q6:  # fp6 corrections: the host supplies a bound of the operand (ops.conv raises without one)
        assert prologue == L.PRO_GN_SILU and pw.act_bound > 0.0 and shortcut is None and tuple(stride) == (1, 1, 1)
        Q6_CALLS.append((float(a.abs().max()), pw.act_bound))
    if pw.batch_stride:
        assert pw.k == (1, 1, 1) and pw.w.shape[0] == B
        y = torch.einsum("bthwc,boc->bthwo", a, pw.w)
    else:
        f = a.permute(0, 4, 1, 2, 3)
        if upsample2x:
            f = F.interpolate(f, scale_factor=(1.0, 2.0, 2.0), mode="nearest")
        f = _pad3(f, pad, pad_mode_t, pad_mode_hw)
        y = F.conv3d(f, pw.w.reshape(pw.cout, pw.cin_real, *pw.k), None, stride=stride).permute(0, 2, 3, 4, 1)
    y = y * alpha + (pw.bias if bias is None else bias)[:pw.cout]
    if shortcut is not None:
        x2, pw2 = shortcut
        y = y + x2.float()[..., :pw2.cin_real] @ pw2.w[:, :, 0].t()
    if residual is not None:
        y = y + residual.float()
    odt = torch.float32 if out_f32 else x.dtype
    cst = pw.cout
    if out_mode == L.OUT_NCDHW:
        res = y.permute(0, 4, 1, 2, 3).contiguous().to(odt)
    elif out_mode == L.OUT_TIME_SHUFFLE:  # "b (n c) t h w -> b c (t n) h w", then frame 0 dropped (vae_blocks3d_sd3.py:358-362)
        cst = pw.cout // 2
        Bo, To, Ho, Wo, _ = y.shape
        res = y.reshape(Bo, To, Ho, Wo, 2, cst).permute(0, 1, 4, 2, 3, 5).reshape(Bo, 2 * To, Ho, Wo, cst)[:, 1:].contiguous().to(odt)
    else:
        cp = pw.cout if cout_pad is None else cout_pad
        res = torch.zeros(*y.shape[:-1], cp, dtype=odt)
        res[..., :pw.cout] = y.to(odt)
    if gn_out:
        kt1 = pw.k[0] == 1 and out_mode == L.OUT_NDHWC and not upsample2x
        return res, FakePart(res, B, cst, gn_out, frames=res.shape[1] if kt1 else 0, slabs=res.shape[1] * 4 if kt1 else 0)
    return res


def gn_silu_apply(x, gn, silu=True, per_frame=False):
    B, T, H, W, C = x.shape
    sc, sh = gn
    shape = (B, T, 1, 1, C) if per_frame else (B, 1, 1, 1, C)
    a = x.float() * sc.reshape(shape) + sh.reshape(shape)
    return (F.silu(a) if silu else a).to(x.dtype)


def layernorm(x, gamma, beta, eps):
    return F.layer_norm(x.float(), (x.shape[-1],), gamma, beta, eps).to(x.dtype)


def attention_d512(q, k, vt, n, scale):
    """softmax(q k^T * scale) v per batch item; probabilities rounded to the storage dtype before the second product"""
    s = torch.einsum("bnd,bmd->bnm", q.float(), k.float()) * scale
    p = torch.softmax(s, -1).to(q.dtype).float()
    return torch.einsum("bnm,bdm->bnd", p, vt.float()[:, :, :n]).to(q.dtype)


def temporal_attention(q, k, v):
    """[B,T,H,W,C]: softmax(q k^T / sqrt(C)) v over the T frames of every pixel"""
    qf, kf, vf = (t.float().permute(0, 2, 3, 1, 4) for t in (q, k, v))  # [B,H,W,T,C]
    p = torch.softmax(qf @ kf.transpose(-1, -2) * (q.shape[-1] ** -0.5), -1)
    return (p @ vf).permute(0, 3, 1, 2, 4).contiguous().to(q.dtype)


def ndhwc_to_ncdhw(x, c):
    return x[..., :c].permute(0, 4, 1, 2, 3).contiguous()


def frames_u8_to_ndhwc(frames, cpad, dtype):
    """uint8 [T,H,W,3] -> [1,T,H,W,cpad]: u8 -> dtype, / 127.5, - 1.0 with the scripts' rounding steps (cvvae_inference_video.py:34)"""
    T, H, W, _ = frames.shape
    out = torch.zeros(1, T, H, W, cpad, dtype=dtype)
    out[0, ..., :3] = frames.to(dtype) / 127.5 - 1.0
    return out


def resize_frames_u8(frames, size):
    """[T,H,W,C] uint8 -> [T,oh,ow,C]: torch's own uint8 antialiased bilinear kernel (what transforms.Resize runs)"""
    return F.interpolate(frames.permute(0, 3, 1, 2), size=tuple(size), mode="bilinear", antialias=True).permute(0, 2, 3, 1).contiguous()


def ncdhw_to_frames_u8(x):
    """[1,3,T,H,W] -> uint8 [T,H,W,3] = u8((clamp(x, -1, 1) + 1) * 127.5) (cvvae_inference_video.py:47-50)"""
    return ((torch.clamp(x[0], -1.0, 1.0) + 1.0) * 127.5).to(torch.uint8).permute(1, 2, 3, 0).contiguous()


def blend_(a, b, overlap, axis):
    """in place on b (NCDHW): b[.., :o] = (1 - w) * a[.., -o:] + w * b[.., :o], w = arange(o) / o in fp32 (modeling_vae.py:321-341)"""
    w = torch.arange(overlap, dtype=torch.float32) / overlap
    if axis == 0:
        w = w[:, None]
        b[:, :, :, :overlap] = ((1 - w) * a[:, :, :, -overlap:].float() + w * b[:, :, :, :overlap].float()).to(b.dtype)
    else:
        b[..., :overlap] = ((1 - w) * a[..., -overlap:].float() + w * b[..., :overlap].float()).to(b.dtype)
    return b


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


def conv_wgrad(a, gy, k, *, stride=(1, 1, 1), pad=((0, 0), (0, 0), (0, 0)), pad_mode_t=L.PAD_ZERO, pad_mode_hw=L.PAD_ZERO, cin=None,
               cout=None, bias=False):
    """dW of conv(a, W) given gy (cvvae_conv_wgrad): autograd of the plain convolution over the padded operand"""
    cin = a.shape[-1] if cin is None else cin
    cout = gy.shape[-1] if cout is None else cout
    f = _pad3(a.float()[..., :cin].permute(0, 4, 1, 2, 3), pad, pad_mode_t, pad_mode_hw)
    w = torch.zeros(cout, cin, *k, requires_grad=True)
    with torch.enable_grad():
        y = F.conv3d(f, w, None, stride=stride)
        assert tuple(y.shape[2:]) == tuple(gy.shape[1:4]), (tuple(y.shape), tuple(gy.shape))
        (y * gy.float()[..., :cout].permute(0, 4, 1, 2, 3)).sum().backward()
    if bias:
        return w.grad.detach(), bias_grad(gy, cout=cout)
    return w.grad.detach()


def bias_grad(gy, cout=None):
    C = gy.shape[-1]
    return gy.float().reshape(-1, C).sum(0)[:C if cout is None else cout]


def gn_bwd_params(x, gy, tabs, gamma, beta, silu, per_frame=False):
    B, T, H, W, C = x.shape
    rows, S = (B * T, H * W) if per_frame else (B, T * H * W)
    rs, nm = tabs
    xh = x.float().reshape(rows, S, C) * rs[:, None, :] + nm[:, None, :]
    a = xh * gamma + beta
    d = torch.sigmoid(a) * (1 + a * (1 - torch.sigmoid(a))) if silu else torch.ones_like(a)
    ga = gy.float().reshape(rows, S, C) * d
    return (ga * xh).sum((0, 1)), ga.sum((0, 1))


def gn_bwd_input_params(x, gy, tabs, gamma, beta, silu, add=None, per_frame=False, groups=32):
    dg, db = gn_bwd_params(x, gy, tabs, gamma, beta, silu, per_frame=per_frame)
    return gn_bwd_input(x, gy, tabs, gamma, beta, silu, add=add, per_frame=per_frame, groups=groups), dg, db


def pad_fold(gp, pad_t, pad_hw, pad_mode_t, pad_mode_hw, add=None):
    """adjoint of _pad3 (cvvae_pad_fold): autograd of the padding itself"""
    B, Tp, Hp, Wp, C = gp.shape
    T, H, W = Tp - pad_t[0] - pad_t[1], Hp - 2 * pad_hw, Wp - 2 * pad_hw
    x = torch.zeros(B, C, T, H, W, requires_grad=True)
    with torch.enable_grad():
        y = _pad3(x, (pad_t, (pad_hw, pad_hw), (pad_hw, pad_hw)), pad_mode_t, pad_mode_hw)
        (y * gp.float().permute(0, 4, 1, 2, 3)).sum().backward()
    out = x.grad.detach().permute(0, 2, 3, 4, 1)
    if add is not None:
        out = out + add.float()
    return out.contiguous().to(gp.dtype)


def temporal_attention_bwd(q, k, v, go):
    B, T, H, W, C = q.shape
    t = [a.float().permute(0, 2, 3, 1, 4).reshape(-1, T, C).clone().requires_grad_(True) for a in (q, k, v)]
    with torch.enable_grad():
        p = torch.softmax(t[0] @ t[1].transpose(1, 2) * (C ** -0.5), -1)
        (p @ t[2] * go.float().permute(0, 2, 3, 1, 4).reshape(-1, T, C)).sum().backward()
    return tuple(a.grad.reshape(B, H, W, T, C).permute(0, 3, 1, 2, 4).contiguous().to(q.dtype) for a in t)


def layernorm_bwd(x, gy, gamma, beta, eps):
    xr = x.float().clone().requires_grad_(True)
    g, b = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    with torch.enable_grad():
        (F.layer_norm(xr, (x.shape[-1],), g, b, eps) * gy.float()).sum().backward()
    return xr.grad.to(x.dtype), g.grad, b.grad


def softmax_bwd_rows(p, gp, n_valid, alpha, ld_o=None):
    rows, ld_p = p.shape
    out = torch.zeros(rows, ld_p if ld_o is None else ld_o, dtype=p.dtype)
    pf, g = p.float()[:, :n_valid], gp[:, :n_valid]
    out[:, :n_valid] = (alpha * pf * (g - (pf * g).sum(-1, keepdim=True))).to(p.dtype)
    return out


def upsample2x_sum(g):
    N, _, H2, W2, C = g.shape
    return g.float().reshape(N, 1, H2 // 2, 2, W2 // 2, 2, C).sum((3, 5)).to(g.dtype)


_NAMES = ["ncdhw_to_rowpack", "ndhwc_to_rowpack", "pack_weight_rowpack", "pack_weight_tapsn", "conv_out_gather", "pack_weight", "pack_weight_tfolds", "pack_weight_t1", "pack_weight_upfold", "pack_weight_batched", "gn_stats",
          "gn_finalize", "gn_silu_apply", "conv", "softmax_rows", "transpose", "layernorm", "attention_d512", "temporal_attention", "ncdhw_to_ndhwc",
          "ndhwc_to_ncdhw", "blend_", "resize_frames_u8", "frames_u8_to_ndhwc", "ncdhw_to_frames_u8", "gn_bwd_input", "softmax_bwd_rows", "upsample2x_sum",
          "conv_wgrad", "bias_grad", "gn_bwd_params", "gn_bwd_input_params", "pad_fold", "temporal_attention_bwd", "layernorm_bwd"]


@contextlib.contextmanager
def patched(whole_model: bool = False):
    """ops.* replaced by the emulations; whole_model: also lets the wrapper classes (modeling._Net / the tile blends) accept CPU
    tensors -- their `is_cuda` guards and torch.cuda.device contexts are the only things between them and the emulated ops."""
    from cvvae_amd import modeling
    saved = {n: getattr(ops, n) for n in _NAMES}
    saved_check, saved_dev, saved_blend = modeling._Net._check_input, torch.cuda.device, modeling._CVVAEBase.__dict__["_blend"]
    try:
        for n in _NAMES:
            setattr(ops, n, globals()[n])
        if whole_model:
            modeling._Net._check_input = lambda self, x: None
            torch.cuda.device = lambda *_a, **_k: contextlib.nullcontext()

            def _blend(a, b, o, axis):
                o = min(a.shape[3 + axis], b.shape[3 + axis], o)  # (mirrors modeling._CVVAEBase._blend without its is_cuda guard)
                if o <= 0:
                    return b
                if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
                    from cvvae_amd.grad3d import BlendFn
                    return BlendFn.apply(a, b, o, axis)
                bc = b.contiguous()
                ops.blend_(a.contiguous(), bc, o, axis)
                if bc is not b:
                    b.copy_(bc)
                return b
            modeling._CVVAEBase._blend = staticmethod(_blend)
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
        modeling._Net._check_input = saved_check
        torch.cuda.device = saved_dev
        modeling._CVVAEBase._blend = saved_blend
