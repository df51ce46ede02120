"""Input gradients through the FROZEN 2-D constraint decoder on the MI355X kernels (training-side codec use).

The reference trains the 3-D VAE with a latent-compatibility loss: the latents go through the frozen SD3 image decoder
(`self.constraint_decoder.requires_grad_(False)`, `xrec_2d = self.constraint_decoder(z)`, lvdm/models/autoencoder.py:1057-1069) and
the loss back-propagates THROUGH that decoder into z (and on into the encoder).  What the frozen module contributes to the
backward pass is autograd's input gradient of each of its ops -- no weight gradients.  This file is that pass for
`engine.constraint_decoder2d`:

  conv / linear   grad_input = conv(gy, W with taps flipped and Cin/Cout exchanged), same zero padding  -> the forward MFMA kernel
                  (WeightCache.conv_dgrad; nn.Conv2d(3x3, padding=1) and nn.Linear of vae_blocks_sd3.py / diffusers Attention)
  GroupNorm+SiLU  cvvae_gn_bwd_input (two deterministic passes), the residual skip summed in the same launch
  upsample        Upsample2D = nearest x2 + conv: the conv's input gradient at the upsampled size, then cvvae_upsample2x_sum
  attention       softmax(QK^T/sqrt(C))V per frame: five more products on the 1x1 kernel with per-frame "weights" packed from
                  activations (as the forward does) + cvvae_softmax_bwd_rows

Gradients are carried in the module's dtype with fp32 accumulation inside every kernel (what autocast training does).
The decoder's own parameters must be frozen; weight gradients (training the codec itself) are not built.
"""
from typing import List, Optional

import torch

from . import _lib as L
from . import engine, ops
from .engine import G32, P2D, ZERO, WeightCache

K2D = (1, 3, 3)
K1 = (1, 1, 1)


def _unit_tabs(wc: WeightCache, x: torch.Tensor, part, eps: float, per_frame: bool = False):
    """(rstd, -mean*rstd) tables [rows, C] of the GroupNorm over x: the forward's statistics with gamma 1, beta 0"""
    C = x.shape[-1]
    one = torch.ones(C, dtype=torch.float32, device=x.device)
    zero = torch.zeros(C, dtype=torch.float32, device=x.device)
    if part is not None and not per_frame:
        return ops.gn_finalize(part, one, zero, eps)
    return ops.gn_stats(x, one, zero, eps, per_frame=per_frame)


def _dgrad3x3(wc: WeightCache, g: torch.Tensor, pre: str, cin_pad=None, **kw) -> torch.Tensor:
    return ops.conv(g, wc.conv_dgrad(pre, K2D, cin_pad=cin_pad), pad=P2D, pad_mode_hw=ZERO, **kw)


def _dgrad1x1(wc: WeightCache, g: torch.Tensor, pre: str, residual=None) -> torch.Tensor:
    """g [..., Cout] -> g . W  ([..., Cin]) on the flattened pixels of every batch row (+ residual)"""
    pw = wc.conv_dgrad(pre, K1)
    y = ops.conv(engine._flat(g), pw, residual=engine._flat(residual) if residual is not None else None)
    return y.view(*g.shape[:-1], pw.cout)


def resnet_backward(wc: WeightCache, g: torch.Tensor, e: dict) -> torch.Tensor:
    """ResnetBlock2D (vae_blocks_sd3.py:368-421): y = conv2(silu(norm2(h))) + shortcut(x), h = conv1(silu(norm1(x))).  g = dL/dy."""
    pre = e["pre"]
    x, h = e["x"], e["h"]
    g_a2 = _dgrad3x3(wc, g, pre + ".conv2")
    g_h = ops.gn_bwd_input(h, g_a2, _unit_tabs(wc, h, e["hp"], 1e-6), *wc.norm(pre + ".norm2"), silu=True)
    g_a1 = _dgrad3x3(wc, g_h, pre + ".conv1")
    skip = _dgrad1x1(wc, g, pre + ".conv_shortcut") if wc.has(pre + ".conv_shortcut.weight") else g
    return ops.gn_bwd_input(x, g_a1, _unit_tabs(wc, x, e["xp"], 1e-6), *wc.norm(pre + ".norm1"), silu=True, add=skip)


def _linear_grads(wc: WeightCache, grads: dict, pre: str, a: torch.Tensor, g: torch.Tensor):
    """parameter gradients of y = linear(a) (nn.Linear / 1x1 conv `pre`) given g = dL/dy: dW = g^T a on the wgrad kernel, db = sum g"""
    w = wc.p(pre + ".weight")
    a5, g5 = a.reshape(a.shape[0], 1, 1, -1, a.shape[-1]), g.reshape(g.shape[0], 1, 1, -1, g.shape[-1])
    grads[pre + ".weight"] = ops.conv_wgrad(a5.contiguous(), g5.contiguous(), K1, cin=w.shape[1], cout=w.shape[0]).reshape(w.shape)
    if wc.has(pre + ".bias"):
        grads[pre + ".bias"] = ops.bias_grad(g5.contiguous(), cout=w.shape[0])


def attention_backward(wc: WeightCache, g: torch.Tensor, e: dict, grads: Optional[dict] = None,
                       add_extra: Optional[torch.Tensor] = None) -> torch.Tensor:
    """engine.spatial_attention: out = x + proj(softmax(q k^T / sqrt(C)) v), q,k,v = linear(GroupNorm(x)) per frame.  g = dL/dout.
    grads: a dict that receives the block's PARAMETER gradients (training the network itself, grad3d.py); None = frozen module."""
    x, qq, kk, vv, p = e["x"], e["qq"], e["kk"], e["vv"], e["p"]
    norm, q, k, v, proj = e["names"]
    B, T, H, W, C = x.shape
    N, BT = H * W, B * T
    npad = p.shape[-1]
    scale = float(C) ** -0.5
    g_o = _dgrad1x1(wc, g, proj).view(BT, 1, 1, N, C)                                            # dL/d(PV)
    # dL/dP[n,m] = sum_c g_o[n,c] V[m,c]  (fp32), then through the softmax and the score scale
    vw = ops.pack_weight_batched(vv.view(BT, N, C), K1, cin_pad=C, strides=(C, 1, 0), cout=N, cin=C)
    g_p = ops.conv(g_o, vw, out_f32=True, cout_pad=npad)                                          # [BT,1,1,N,npad]
    g_s = ops.softmax_bwd_rows(p.view(BT * N, npad), g_p.view(BT * N, npad), N, scale)            # [BT*N, npad]
    # dL/dV[m,c] = sum_n P[n,m] g_o[n,c]
    p_t = ops.transpose(p.view(BT, N, npad), ncols=N, ld_out=npad)                                # [BT, N(m), npad(n)]
    gow = ops.pack_weight_batched(ops.transpose(g_o.view(BT, N, C)), K1, cin_pad=npad, strides=(N, 1, 0), cout=C, cin=N)
    g_v = ops.conv(p_t.view(BT, 1, 1, N, npad), gow)                                              # [BT,1,1,N,C]
    # dL/dQ[n,c] = sum_m g_s[n,m] K[m,c]
    kw = ops.pack_weight_batched(ops.transpose(kk.view(BT, N, C)), K1, cin_pad=npad, strides=(N, 1, 0), cout=C, cin=N)
    g_q = ops.conv(g_s.view(BT, 1, 1, N, npad), kw)
    # dL/dK[m,c] = sum_n g_s[n,m] Q[n,c]
    gs_t = ops.transpose(g_s.view(BT, N, npad), ncols=N, ld_out=npad)
    qw = ops.pack_weight_batched(ops.transpose(qq.view(BT, N, C)), K1, cin_pad=npad, strides=(N, 1, 0), cout=C, cin=N)
    g_k = ops.conv(gs_t.view(BT, 1, 1, N, npad), qw)
    # back through to_q / to_k / to_v into the normalised input, summed in the launches' residual inputs
    g_n = _dgrad1x1(wc, g_q, q)
    g_n = _dgrad1x1(wc, g_k, k, residual=g_n)
    g_n = _dgrad1x1(wc, g_v, v, residual=g_n)
    tabs = _unit_tabs(wc, x, None, e["eps"], per_frame=True)
    if grads is not None:
        n = ops.gn_silu_apply(x, e["gn"], silu=False, per_frame=True).view(BT, 1, 1, N, C)   # what to_q / to_k / to_v consumed
        _linear_grads(wc, grads, proj, e["o"].view(BT, 1, 1, N, C), g.view(BT, 1, 1, N, C))
        _linear_grads(wc, grads, q, n, g_q)
        _linear_grads(wc, grads, k, n, g_k)
        _linear_grads(wc, grads, v, n, g_v)
        # (add_extra: a block without its own residual -- vae3d's spatial-temporal attention -- passes the gradient of the outer one)
        gx, grads[norm + ".weight"], grads[norm + ".bias"] = ops.gn_bwd_input_params(
            x, g_n.view(B, T, H, W, C), tabs, *wc.norm(norm), silu=False, add=g if e["residual"] else add_extra, per_frame=True)
        return gx
    return ops.gn_bwd_input(x, g_n.view(B, T, H, W, C), tabs, *wc.norm(norm), silu=False, add=g if e["residual"] else add_extra,
                            per_frame=True)


def constraint_decoder2d_backward(wc: WeightCache, tape: List[dict], gy: torch.Tensor) -> torch.Tensor:
    """gy = dL/d(output) [b,3,t,H,W] of engine.constraint_decoder2d(wc, z, cfg, tape) -> dL/dz [b,c,t,h,w]."""
    dtype = wc.p("conv_in.weight").dtype
    last = tape[-1]
    assert last["op"] == "out"
    B, T, zin = last["B"], last["T"], last["zin"]
    g = ops.ncdhw_to_ndhwc(gy.contiguous(), 32, dtype)                                           # [b,t,H,W,32], channels 3.. zero
    g = g.view(B * T, 1, g.shape[2], g.shape[3], 32)
    g = _dgrad3x3(wc, g, "conv_out", cin_pad=32)                                                 # dL/d silu(norm_out(h))
    x = last["x"]
    g = ops.gn_bwd_input(x, g, _unit_tabs(wc, x, last["xp"], 1e-6), *wc.norm("conv_norm_out"), silu=True)
    for e in reversed(tape[:-1]):
        if e["op"] == "resnet":
            g = resnet_backward(wc, g, e)
        elif e["op"] == "attn":
            g = attention_backward(wc, g, e)
        elif e["op"] == "up":  # Upsample2D: nearest x2, then conv 3x3
            g = ops.upsample2x_sum(_dgrad3x3(wc, g, e["pre"]))
        else:
            raise AssertionError(e["op"])
    gz = ops.conv(g, wc.conv_dgrad("conv_in", K2D), pad=P2D, pad_mode_hw=ZERO, out_mode=L.OUT_NCDHW)   # [b*t, zin, 1, h, w]
    return gz.view(B, T, zin, gz.shape[3], gz.shape[4]).transpose(1, 2).contiguous()


class ConstraintDecoderFn(torch.autograd.Function):
    """z -> Decoder(z) with the frozen decoder's input gradient as backward (both on the HIP kernels)"""

    @staticmethod
    def forward(ctx, z: torch.Tensor, net) -> torch.Tensor:
        tape: List[dict] = []
        with torch.cuda.device(z.device):
            y = engine.constraint_decoder2d(net._cache(), z.detach(), net._cfg, tape)
        ctx.net, ctx.tape, ctx.zdtype = net, tape, z.dtype
        ctx.cd = net._cache().compute_dtype  # (torch.autocast: the backward thread runs outside the context -- same weights, same dtype)
        return y

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        net, tape = ctx.net, ctx.tape
        # the tape (block inputs, statistics records, attention operands) is kept until autograd frees the node, so a second
        # backward through it (retain_graph=True, two losses) walks the same tape and gives the same bits
        with torch.cuda.device(gy.device), net._cache().computing_in(ctx.cd):
            gz = constraint_decoder2d_backward(net._cache(), tape, gy)
        return gz.to(ctx.zdtype), None
