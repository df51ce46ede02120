"""Backward pass of the TRAINABLE 3-D encoder on the MI355X kernels (second half).

The reference's training step runs the 3-D VAE itself under autograd -- `z, xrec, reg = self(x)`, then the latent-compatibility loss
through the frozen 2-D decoder (`xrec_2d = self.constraint_decoder(z)`; /root/reference/lvdm/models/autoencoder.py:1057-1090) -- so
every layer of `Encoder3D` / `Decoder3D` (/root/reference/models/vae_models3d_sd3.py:162-208, 323-388) owes autograd its input
gradient AND its parameter gradients.  `grad.py` is the frozen half (input gradients only); this file is the trainable half for the
vae3d_sd3 family:

  conv 3x3x3, replicate padding (CausalConv3d T(2,0) / Conv3d T(1,1); vae_blocks3d_sd3.py:16-104)
      input gradient   the FULL correlation of gy with the tap-flipped, transposed weights (the forward MFMA kernel, zero pad 2 on
                       every side = the gradient w.r.t. the PADDED input), then `cvvae_pad_fold`: the adjoint of the replicate
                       coordinate map (border elements collect their pad region, e.g. frame 0 its two causal copies)
      strided          Downsample3D (stride (2,2,2) / (1,2,2)): gy is zero-stuffed onto the stride-1 grid first (4-8x wasted MFMAs on
                       three small layers; a first version)
      weight gradient  `cvvae_conv_wgrad` over the operand the forward multiplied (GroupNorm + SiLU re-applied by
                       `cvvae_gn_silu_apply`; padding by the kernel's own coordinate map), bias gradient `cvvae_channel_sums`
  conv 1x3x3 zero padding, 1x1 shortcut, nn.Linear   as in grad.py + the same wgrad / bias kernels
  GroupNorm (+ SiLU)  input gradient `cvvae_gn_bwd_input`, affine gradients `cvvae_channel_sums` (both from the forward's statistics)
  attention           grad.attention_backward with its parameter gradients switched on

  Upsample3D          (decoder) the adjoint of the time shuffle + frame drop (index plumbing), the 27-tap conv's gradients over the
                      nearest-upsampled operand, then `cvvae_upsample2x_sum`

`run_trainable` makes a taped forward + this backward TWO autograd nodes over the clip (or latent) and the module's parameters --
`Net3DBodyFn` (everything up to the last GroupNorm's input) and `Net3DTailFn` (GroupNorm + SiLU + conv_out: what
`torch.autograd.grad(loss, get_last_layer())` of the reference's adversarial loss needs, without the body's backward) --
so `loss(x, decoder(encoder(x)), constraint_decoder(z)).backward()` fills `<net>.<param>.grad` (modeling._Net.forward takes this path
for a module in train() mode under grad mode).  Gradients are carried in the module's dtype with fp32 accumulation inside every
kernel; parameter gradients are accumulated in fp32 and cast to the parameter's dtype at the end.  The walker takes every layer's
padding, GroupNorm eps and names from the tape, so the vae3d (SD2.1-compatible) family's ENCODER (vae_models.py:790-823: zero H / W
padding, asymmetric Downsample3D pads, eps 1e-5) trains through it too, and so does its DECODER (vae_models.py:960-1002): the temporal half of MemoryEfficientAttnVideoBlock
has its own backward kernel (`cvvae_temporal_attention_bwd`), its LayerNorm runs on the GroupNorm kernels (one group, rows = tokens).
Not built yet: spatial tiling / temporal windows under autograd (one window, one tile per call: the training crops of the
reference's configs fit one).
"""
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from . import engine, grad, ops
from .engine import P2D, REP, ZERO, WeightCache

K333 = (3, 3, 3)
K133 = (1, 3, 3)
K1 = (1, 1, 1)
FULL = ((2, 2), (2, 2), (2, 2))  # zero padding of the full correlation of a 3x3x3 kernel


def _conv_param_grads(wc: WeightCache, grads: Optional[Dict[str, torch.Tensor]], pre: str, a, g: torch.Tensor, k, **geom):
    """dW, db of `pre` (a conv over operand a with output gradient g); grads None = a frozen network: nothing to do.
    a: the operand, or a callable that produces it (so that a frozen pass does not re-create operands it never reads)"""
    if grads is None:
        return
    if callable(a):
        a = a()
    w = wc.p(pre + ".weight")
    if wc.has(pre + ".bias"):   # (the bias gradient comes out of the weight-gradient launch where the kernel fuses it)
        dw, grads[pre + ".bias"] = ops.conv_wgrad(a, g, k, cin=w.shape[1], cout=w.shape[0], bias=True, **geom)
    else:
        dw = ops.conv_wgrad(a, g, k, cin=w.shape[1], cout=w.shape[0], **geom)
    grads[pre + ".weight"] = dw.reshape(w.shape)


def _gn_backward(grads, name: str, x, g, tabs, affine, silu: bool, add=None):
    """input gradient of act(GroupNorm(x)) (+ add), and -- trainable network -- the norm's affine gradients into `grads`"""
    if grads is None:
        return ops.gn_bwd_input(x, g, tabs, *affine, silu=silu, add=add)
    gx, grads[name + ".weight"], grads[name + ".bias"] = ops.gn_bwd_input_params(x, g, tabs, *affine, silu=silu, add=add)
    return gx


def dgrad333(wc: WeightCache, g: torch.Tensor, pre: str, pad, mode_t: int, mode_hw: int, in_shape, stride=(1, 1, 1),
             add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """input gradient of a 3x3x3 conv with padding `pad` = ((tf, tb), (hf, hb), (wf, wb)) in modes (mode_t, mode_hw) and `stride`:
    g [B,To,Ho,Wo,Cout] -> [B,T,H,W,Cin] (+ add).  The full correlation with the tap-flipped, transposed weights gives the gradient
    w.r.t. the PADDED input; the adjoint of the padding folds it back (replicate: cvvae_pad_fold; zero: a crop)."""
    B, T, H, W = in_shape
    (tf, tb), (hf, hb), (wf, wb) = pad
    Tz, Hz, Wz = T + tf + tb - 2, H + hf + hb - 2, W + wf + wb - 2   # the stride-1 output grid = padded extent - 2 per axis
    if tuple(stride) != (1, 1, 1):
        gz = g.new_zeros((B, Tz, Hz, Wz, g.shape[-1]))                # zero-stuffed: positions o*s carry g
        gz[:, ::stride[0], ::stride[1], ::stride[2]][:, :g.shape[1], :g.shape[2], :g.shape[3]] = g
        g = gz
    assert tuple(g.shape[1:4]) == (Tz, Hz, Wz), (tuple(g.shape), in_shape, pad)
    pw = wc.conv_dgrad(pre, K333)
    cp = ops.round_up(pw.cout, 8)  # (conv_in: 3 input channels -> an 8-channel gradient tensor, pad channels zero)
    gp = ops.conv(g, pw, pad=FULL, pad_mode_t=ZERO, pad_mode_hw=ZERO, cout_pad=cp if cp != pw.cout else None)  # [B,T+tf+tb,H+hf+hb,W+wf+wb,Cin]
    if mode_hw == ZERO:   # zero padding in H / W: the adjoint is the interior crop (any front / back split)
        if hf or hb or wf or wb:
            gp = gp[:, :, hf:hf + H, wf:wf + W].contiguous()
        phw = 0
    else:
        assert hf == hb == wf == wb, "replicate H / W padding: symmetric"
        phw = hf
    return ops.pad_fold(gp, (tf, tb), phw, mode_t, mode_hw, add=add)


def dgrad333_replicate(wc, g, pre, pad_t, in_shape, stride=(1, 1, 1), add=None):
    """(the vae3d_sd3 flavour: replicate everywhere, H / W pads 1)"""
    return dgrad333(wc, g, pre, (tuple(pad_t), (1, 1), (1, 1)), REP, REP, in_shape, stride=stride, add=add)


def sd3_resnet_backward(wc: WeightCache, g: torch.Tensor, e: dict, grads: Dict[str, torch.Tensor]) -> torch.Tensor:
    """ResnetBlock3D of either family (vae_blocks3d_sd3.py:517-569, vae_models.py:390-410): y = conv2(silu(norm2(h))) + shortcut(x),
    h = conv1(silu(norm1(x))); g = dL/dy.  Padding, GroupNorm eps and the shortcut's name come with the tape entry."""
    pre, x, h = e["pre"], e["x"], e["h"]
    pad, mt, mhw, eps = e["pad"], e["mode_t"], e["mode_hw"], e["eps"]
    B, T, H, W, _ = x.shape
    # conv2: per-frame 3x3, zero padding, over a2 = silu(norm2(h))
    _conv_param_grads(wc, grads, pre + ".conv2", lambda: ops.gn_silu_apply(h, e["g2"]), g, K133, pad=P2D)
    g_a2 = ops.conv(g, wc.conv_dgrad(pre + ".conv2", K133), pad=P2D, pad_mode_hw=ZERO)
    tabs2 = grad._unit_tabs(wc, h, e["hp"], eps)
    n2 = wc.norm(pre + ".norm2")
    g_h = _gn_backward(grads, pre + ".norm2", h, g_a2, tabs2, n2, True)
    del g_a2
    # conv1: 3x3x3 over a1 = silu(norm1(x))
    _conv_param_grads(wc, grads, pre + ".conv1", lambda: ops.gn_silu_apply(x, e["g1"]), g_h, K333, pad=pad, pad_mode_t=mt,
                      pad_mode_hw=mhw)
    g_a1 = dgrad333(wc, g_h, pre + ".conv1", pad, mt, mhw, (B, T, H, W))
    # skip branch
    sc = pre + e["sc"]
    if wc.has(sc + ".weight"):
        if grads is not None:
            grad._linear_grads(wc, grads, sc, x.view(B, 1, 1, -1, x.shape[-1]), g.view(B, 1, 1, -1, g.shape[-1]))
        skip = grad._dgrad1x1(wc, g, sc)
    else:
        skip = g
    tabs1 = grad._unit_tabs(wc, x, e["xp"], eps)
    n1 = wc.norm(pre + ".norm1")
    return _gn_backward(grads, pre + ".norm1", x, g_a1, tabs1, n1, True, add=skip)


def _unshuffle_time(g: torch.Tensor) -> torch.Tensor:
    """adjoint of Upsample3D's 'b (n c) t h w -> b c (t n) h w' + drop of frame 0 (vae_blocks3d_sd3.py:358-362): g [B,2T-1,H,W,C]
    (stored frames 2t+n-1) -> the conv output's gradient [B,T,H,W,2C] (channel n*C+c of frame t; the dropped frame gets zeros)"""
    B, F2, H, W, C = g.shape
    gf = torch.cat([g.new_zeros((B, 1, H, W, C)), g], dim=1)                      # frame -1 back in place
    return gf.view(B, (F2 + 1) // 2, 2, H, W, C).permute(0, 1, 3, 4, 2, 5).reshape(B, (F2 + 1) // 2, H, W, 2 * C).contiguous()


def sd3_upsample_backward(wc: WeightCache, g: torch.Tensor, e: dict, grads: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Upsample3D (vae_blocks3d_sd3.py:314-364): nearest x(1,2,2) -> 3x3x3 conv (replicate pad) -> time shuffle + drop.  The
    forward ran the folded 3x2x2 phase form; the backward differentiates the op it equals: the 27-tap conv over the upsampled
    operand (materialised here -- a first version), then the 2x2 block sum of the nearest upsample's adjoint."""
    x, pre, pad = e["x"], e["pre"], e["pad"]
    B, T, H, W, C = x.shape
    gc = _unshuffle_time(g) if e["up_time"] else g
    # (the operand: F.interpolate(scale=(1,2,2), mode="nearest") of x, materialised for the weight gradient only)
    _conv_param_grads(wc, grads, pre, lambda: x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3), gc, K333, pad=pad,
                      pad_mode_t=e["mode_t"], pad_mode_hw=e["mode_hw"])
    gup = dgrad333(wc, gc, pre, pad, e["mode_t"], e["mode_hw"], (B, T, 2 * H, 2 * W))
    return ops.upsample2x_sum(gup.view(B * T, 1, 2 * H, 2 * W, C)).view(B, T, H, W, C)


def temporal_attention_backward(wc: WeightCache, g: torch.Tensor, e: dict, grads: Dict[str, torch.Tensor]) -> torch.Tensor:
    """the temporal half of MemoryEfficientAttnVideoBlock (vae_models.py:573-587, 619-629): out = x + proj_out_t(attention_t(q_t(n),
    k_t(n), v_t(n))), n = LayerNorm(h), h = the spatial attention's output.  g = dL/dout -> dL/dh (the outer residual's gradient g is
    added by the caller after the spatial half)."""
    a, h, n = e["pre"], e["h"], e["n"]
    B, T, H, W, C = h.shape
    flat = lambda t: t.view(B, 1, 1, -1, t.shape[-1])  # noqa: E731
    if grads is not None:
        grad._linear_grads(wc, grads, a + ".proj_out_t", flat(e["o"]), flat(g))
    g_o = grad._dgrad1x1(wc, g, a + ".proj_out_t")
    g_q, g_k, g_v = ops.temporal_attention_bwd(e["q"], e["k"], e["v"], g_o)
    g_n = None
    for name, gg in ((".q_t", g_q), (".k_t", g_k), (".v_t", g_v)):
        if grads is not None:
            grad._linear_grads(wc, grads, a + name, flat(n), flat(gg))
        g_n = grad._dgrad1x1(wc, gg, a + name, residual=g_n)
    g_h, dg, db = ops.layernorm_bwd(h, g_n, *wc.norm(a + ".norm_t"), 1e-5)
    if grads is not None:
        grads[a + ".norm_t.weight"], grads[a + ".norm_t.bias"] = dg, db
    return g_h


def tail_backward(wc: WeightCache, last: dict, gy: torch.Tensor, grads: Optional[Dict[str, torch.Tensor]], need_input_grad: bool = True):
    """backward of a network's last two layers (the `out3d` tape entry: GroupNorm + SiLU + conv_out): gy = dL/d(output) (NCDHW) ->
    dL/dh for the entry's NDHWC input h (None when not needed and the norm is frozen); conv_out / norm gradients into `grads`"""
    assert last["op"] == "out3d"
    dtype = wc.p("conv_in.weight").dtype
    pad, mt, mhw = last["pad"], last["mode_t"], last["mode_hw"]
    cout = wc.p("conv_out.weight").shape[0]
    g = ops.ncdhw_to_ndhwc(gy.contiguous(), ops.round_up(cout, 16), dtype)                      # [B,T',h,w,Cpad], pad channels zero
    x = last["x"]
    _conv_param_grads(wc, grads, "conv_out", lambda: ops.gn_silu_apply(x, last["g"]), g, K333, pad=pad, pad_mode_t=mt, pad_mode_hw=mhw)
    if not need_input_grad and grads is None:
        return None
    g = dgrad333(wc, g, "conv_out", pad, mt, mhw, tuple(x.shape[:4]))
    tabs = grad._unit_tabs(wc, x, last["xp"], last["eps"])
    no = wc.norm(last["norm"])
    return _gn_backward(grads, last["norm"], x, g, tabs, no, True)


def body_backward(wc: WeightCache, tape: List[dict], g: torch.Tensor, grads: Optional[Dict[str, torch.Tensor]], need_input_grad: bool):
    """the tape without its `out3d` entry, walked backwards from g = dL/dh (NDHWC) -> dL/d(input) NCDHW or None"""
    dtype = wc.p("conv_in.weight").dtype
    gx = None
    outer = None  # gradient of the outer residual of a spatial-temporal attention block, added after its spatial half
    for e in reversed(tape):
        if e["op"] == "resnet3d":
            g = sd3_resnet_backward(wc, g, e, grads)
        elif e["op"] == "attn_t":
            outer, g = g, temporal_attention_backward(wc, g, e, grads)
        elif e["op"] == "attn":
            g = grad.attention_backward(wc, g, e, grads, add_extra=outer)
            outer = None
        elif e["op"] == "down3d":
            xin = e["x"]
            _conv_param_grads(wc, grads, e["pre"], xin, g, K333, stride=e["stride"], pad=e["pad"], pad_mode_t=e["mode_t"],
                              pad_mode_hw=e["mode_hw"])
            g = dgrad333(wc, g, e["pre"], e["pad"], e["mode_t"], e["mode_hw"], tuple(xin.shape[:4]), stride=e["stride"])
        elif e["op"] == "up3d":
            g = sd3_upsample_backward(wc, g, e, grads)
        elif e["op"] == "conv_in":   # the encoder's first layer over the clip
            if grads is None and not need_input_grad:
                continue
            xin = e["x"] if e["ndhwc_in"] else ops.ncdhw_to_ndhwc(e["x"], 16, dtype)           # [B,T,H,W,16], channels 3.. zero
            _conv_param_grads(wc, grads, "conv_in", xin, g, K333, pad=e["pad"], pad_mode_t=e["mode_t"], pad_mode_hw=e["mode_hw"])
            if need_input_grad:
                # (conv_in's weights as a 128 -> 3-channel transposed kernel; the gradient tensor is channel-padded to 8)
                gi = dgrad333(wc, g, "conv_in", e["pad"], e["mode_t"], e["mode_hw"], tuple(xin.shape[:4]))
                gx = ops.ndhwc_to_ncdhw(gi, wc.p("conv_in.weight").shape[1])
        elif e["op"] == "dec_in":    # the decoder's first layer over the (channel-padded NDHWC) latent
            xin = e["x"]
            _conv_param_grads(wc, grads, "conv_in", xin, g, K333, pad=e["pad"], pad_mode_t=e["mode_t"], pad_mode_hw=e["mode_hw"])
            if need_input_grad:
                gi = dgrad333(wc, g, "conv_in", e["pad"], e["mode_t"], e["mode_hw"], tuple(xin.shape[:4]))
                gx = ops.ndhwc_to_ncdhw(gi, e["zin"])
        else:
            raise AssertionError(e["op"])
    return gx


def sd3_net_backward(wc: WeightCache, tape: List[dict], gy: torch.Tensor, need_input_grad: bool = False, need_params: bool = True):
    """gy = dL/d(output) (NCDHW) of engine.sd3_encoder / engine.sd3_decoder run with `tape` -> (dL/d(input) NCDHW or None,
    {parameter name: fp32 gradient}).  need_params = False (a FROZEN network whose input needs a gradient): the weight-gradient,
    bias and affine launches are skipped and the dict comes back empty."""
    grads: Optional[Dict[str, torch.Tensor]] = {} if need_params else None
    g = tail_backward(wc, tape[-1], gy, grads)
    gx = body_backward(wc, tape[:-1], g, grads, need_input_grad)
    return gx, (grads if grads is not None else {})


sd3_encoder_backward = sd3_net_backward  # (the encoder's tape through the common walker)
sd3_decoder_backward = sd3_net_backward


def _grads_out(names, pmeta, grads):
    """the fp32 gradients in the parameters' order, shapes and dtypes.  The conversions of a 16-bit model are ONE multi-tensor copy
    (a `.to(dt)` per parameter was 244 five-microsecond launches per training step of the sd3 pair)."""
    out, src, dst = [], [], []
    for name, (dt, req, shape) in zip(names, pmeta):
        gq = grads.get(name)
        if not (req and gq is not None):
            out.append(None)
            continue
        gq = gq.reshape(shape)
        if gq.dtype != dt:
            src.append(gq)
            gq = torch.empty(shape, dtype=dt, device=gq.device)
            dst.append(gq)
        out.append(gq)
    if dst:
        torch._foreach_copy_(dst, src)
    return out


# A network is TWO autograd nodes: the body (everything up to the input h of the last GroupNorm) and the tail (GroupNorm + SiLU +
# conv_out).  The reference's adversarial loss asks for  torch.autograd.grad(loss, decoder.get_last_layer(), retain_graph=True)  twice
# per step (`calculate_adaptive_weight`, lvdm/modules/autoencoding/losses/discriminator_loss.py:211-220; get_last_layer =
# conv_out.weight, vae_models3d_sd3.py:390-391): with the tail as its own node the engine runs ONLY the tail's backward for those
# calls (one weight gradient, one input gradient, one GroupNorm backward) instead of the whole network's.  The forward is still ONE
# pass of the inference program: the body node runs it, hands the tail's input h out as its output and parks y for the tail node.
TAIL_PREFIXES = ("conv_out", "conv_norm_out", "norm_out")


def _is_tail(name: str) -> bool:
    return name.rsplit(".", 1)[0] in TAIL_PREFIXES


# The backward reads the weights LIVE (packed input-gradient forms, norm affines) instead of saving them on the tape.  PyTorch's own
# conv backward would raise "one of the variables needed for gradient computation has been modified by an inplace operation" when
# a parameter changes between forward and backward (an optimizer.step() under retain_graph, a GAN's generator / discriminator
# alternation on one graph); so does this one: the parameters' (storage, version) are remembered by the forward and checked.
def _remember_versions(ctx, params):
    ctx.pobj = params
    ctx.pver = [(p.data_ptr(), p._version) for p in params]


def _check_unmodified(ctx):
    for name, p, was in zip(ctx.names, ctx.pobj, ctx.pver):
        if (p.data_ptr(), p._version) != was:
            raise RuntimeError(f"parameter {name} of {type(ctx.net).__name__} was modified (in place, or replaced) between the forward "
                               f"and this backward pass: the taped activations belong to the old weights (version {was[1]} -> "
                               f"{p._version}).  Run the backward before optimizer.step(), or re-run the forward.")


class Net3DBodyFn(torch.autograd.Function):
    """(x, *body parameters) -> h, the NDHWC input of the network's last GroupNorm (the whole inference program runs here, with a
    tape; its output y waits in `box` for Net3DTailFn); backward = body_backward"""

    @staticmethod
    def forward(ctx, x: torch.Tensor, net, names: Tuple[str, ...], box: dict, *params) -> torch.Tensor:
        tape: List[dict] = []
        with torch.cuda.device(x.device):
            y = type(net)._program(net._cache(), x.detach(), dict(net._cfg), tape)
        assert tape[-1]["op"] == "out3d" and tape[-1]["norm"] in TAIL_PREFIXES, tape[-1]["op"]
        box["y"], box["last"] = y, tape[-1]
        ctx.net, ctx.tape, ctx.names = net, tape[:-1], names
        ctx.cd = box["cd"] = net._cache().compute_dtype  # (autocast: the backward runs outside the context -- same 16-bit weight copies)
        ctx.x_dtype, ctx.need_x = x.dtype, x.requires_grad
        ctx.pmeta = [(p.dtype, p.requires_grad, tuple(p.shape)) for p in params]
        _remember_versions(ctx, params)
        # (a view: autograd owns the returned tensor object, the tape keeps reading the same storage)
        return tape[-1]["x"].view_as(tape[-1]["x"])

    @staticmethod
    def backward(ctx, gh: torch.Tensor):
        need_params = any(req for _, req, _ in ctx.pmeta)
        grads: Optional[Dict[str, torch.Tensor]] = {} if need_params else None
        _check_unmodified(ctx)
        with torch.cuda.device(gh.device), ctx.net._cache().computing_in(ctx.cd):
            gx = body_backward(ctx.net._cache(), ctx.tape, gh.contiguous(), grads, ctx.need_x)
        return (gx.to(ctx.x_dtype) if gx is not None else None, None, None, None, *_grads_out(ctx.names, ctx.pmeta, grads or {}))


class Net3DTailFn(torch.autograd.Function):
    """(h, *tail parameters) -> y (already computed by the body node's pass); backward = tail_backward"""

    @staticmethod
    def forward(ctx, h: torch.Tensor, net, names: Tuple[str, ...], box: dict, *params) -> torch.Tensor:
        ctx.net, ctx.last, ctx.names, ctx.cd = net, box.pop("last"), names, box.pop("cd")
        ctx.need_h = h.requires_grad
        ctx.pmeta = [(p.dtype, p.requires_grad, tuple(p.shape)) for p in params]
        _remember_versions(ctx, params)
        return box.pop("y")

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        need_params = any(req for _, req, _ in ctx.pmeta)
        grads: Optional[Dict[str, torch.Tensor]] = {} if need_params else None
        _check_unmodified(ctx)
        with torch.cuda.device(gy.device), ctx.net._cache().computing_in(ctx.cd):
            gh = tail_backward(ctx.net._cache(), ctx.last, gy, grads, need_input_grad=ctx.need_h)
        return (gh if ctx.need_h else None, None, None, None, *_grads_out(ctx.names, ctx.pmeta, grads or {}))


class Net3DRecomputeFn(torch.autograd.Function):
    """(x, *parameters) -> y with NOTHING kept but x: the backward re-runs the taped pass and walks it at once -- activation
    recomputation at network-call granularity (the reference wraps every block in torch.utils.checkpoint while training,
    lvdm/common.py:85-104, vae_models3d_sd3.py:167-193, 333-370).  Under the tiled wrapper every (window, tile) unit is one such
    node, so the tapes of the units exist one at a time: peak memory is ONE unit's tape instead of the clip's.  Costs one extra
    forward pass per backward; the last-layer probes (Net3DTailFn's purpose) then cost a whole pass each: `net.recompute` is for
    clips whose tape does not fit, not the default."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, net, names: Tuple[str, ...], *params) -> torch.Tensor:
        with torch.cuda.device(x.device):
            y = type(net)._program(net._cache(), x.detach(), dict(net._cfg))
        ctx.net, ctx.names, ctx.x = net, names, x.detach()
        ctx.cd = net._cache().compute_dtype
        ctx.x_dtype, ctx.need_x = x.dtype, x.requires_grad
        ctx.pmeta = [(p.dtype, p.requires_grad, tuple(p.shape)) for p in params]
        _remember_versions(ctx, params)
        return y

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        _check_unmodified(ctx)
        need_params = any(req for _, req, _ in ctx.pmeta)
        net = ctx.net
        with torch.cuda.device(gy.device), net._cache().computing_in(ctx.cd):
            tape: List[dict] = []
            type(net)._program(net._cache(), ctx.x, dict(net._cfg), tape)  # same launches, same bits as the forward's pass
            gx, grads = sd3_net_backward(net._cache(), tape, gy.contiguous(), need_input_grad=ctx.need_x, need_params=need_params)
            del tape
        return (gx.to(ctx.x_dtype) if gx is not None else None, None, None, *_grads_out(ctx.names, ctx.pmeta, grads))


class BlendFn(torch.autograd.Function):
    """blend_v / blend_h of the tiled wrapper under autograd (modeling_vae.py:321-341): out = b with its first o rows (columns)
    replaced by (1 - w) a[-o:] + w b[:o], w = i / o (the forward kernel on a copy of b); linear, so
    dL/da[-o:] = (1 - w) g[:o], dL/db = g with its first o rows (columns) scaled by w."""

    @staticmethod
    def forward(ctx, a: torch.Tensor, b: torch.Tensor, o: int, axis: int) -> torch.Tensor:
        out = b.detach().contiguous().clone()
        with torch.cuda.device(b.device):
            ops.blend_(a.detach().contiguous(), out, o, axis)
        ctx.o, ctx.axis, ctx.a_shape = o, axis, tuple(a.shape)
        return out

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        o, dim = ctx.o, 3 + ctx.axis
        shape = [1] * 5
        shape[dim] = o
        w = (torch.arange(o, device=g.device, dtype=torch.float32) / float(o)).view(shape)
        head = g.narrow(dim, 0, o).float()
        ga = g.new_zeros(ctx.a_shape)
        ga.narrow(dim, ctx.a_shape[dim] - o, o).copy_(((1.0 - w) * head).to(g.dtype))
        gb = g.clone()
        gb.narrow(dim, 0, o).copy_((w * head).to(g.dtype))
        return ga, gb, None, None


def run_trainable(net, x: torch.Tensor, kwargs: dict) -> torch.Tensor:
    """modeling._Net.forward for a module in train() mode under grad mode: two autograd nodes over (x, parameters) -- body and tail
    (or, `net.recompute`, one node that keeps only x and re-runs the taped pass in its backward)"""
    if kwargs:
        raise NotImplementedError(f"training-mode forward takes no extra arguments (got {sorted(kwargs)})")
    named = [(n, p) for n, p in net.named_parameters()]
    if getattr(net, "recompute", False):
        return Net3DRecomputeFn.apply(x, net, tuple(n for n, _ in named), *[p for _, p in named])
    body = [(n, p) for n, p in named if not _is_tail(n)]
    tail = [(n, p) for n, p in named if _is_tail(n)]
    box: dict = {}
    h = Net3DBodyFn.apply(x, net, tuple(n for n, _ in body), box, *[p for _, p in body])
    return Net3DTailFn.apply(h, net, tuple(n for n, _ in tail), box, *[p for _, p in tail])
