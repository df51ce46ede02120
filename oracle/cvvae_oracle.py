"""CPU oracle for the CV-VAE encode/decode path.  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
file, and there only as the checker.  The product (`cvvae_amd/`) never imports it and fails
loudly when the HIP extension is missing.

What it is: a plain fp32 (optionally fp64) functional restatement, on PyTorch-CPU tensor ops, of the
algorithm in the reference's hot path.  Every function cites the reference
file:line it follows (paths relative to /root/reference).  It consumes the reference's on-disk
state-dict keys directly.

Pinning status: the upstream project ships NO tests / golden vectors for this path ("parity unpinned"
upstream).  This oracle is therefore pinned against outputs of the reference's own, unmodified modules
executed in the build container (oracle/ref_loader.py + oracle/make_golden.py) -- see
tests/golden/*.npz and tests/test_oracle_golden.py, and tests/test_oracle_vs_reference.py which
re-runs the comparison live whenever /root/reference is present.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------------
# shared pieces
# --------------------------------------------------------------------------------------------
def _gn(x, sd: SD, pre: str, eps: float, groups: int = 32):
    """torch.nn.GroupNorm on the tensor as given (5-D: stats over C/G,T,H,W -- Appendix A.5)."""
    return F.group_norm(x, groups, sd[pre + ".weight"], sd[pre + ".bias"], eps)


def _swish(x):
    """models/vae_models.py:187-189 (x*sigmoid(x)) == nn.SiLU (models/vae_blocks3d_sd3.py:482)."""
    return x * torch.sigmoid(x)


def _conv2d_extra_dim(x, w, b, padding: int):
    """Conv2dWithExtraDim: fold T into batch, zero-padded nn.Conv2d, unfold.
    models/vae_blocks3d_sd3.py:107-116, models/vae_models.py:331-340."""
    bsz, c, t, h, ww = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(bsz * t, c, h, ww)
    y = F.conv2d(y, w, b, padding=padding)
    return y.reshape(bsz, t, y.shape[1], h, ww).permute(0, 2, 1, 3, 4)


def _time_shuffle(x, up_time: int):
    """'b (n c) t h w -> b c (t n) h w' then drop frame 0 when up_time == 2.
    models/vae_blocks3d_sd3.py:358-362, models/vae_models.py:230-232 (Appendix A.12)."""
    if up_time == 1:
        return x
    b, nc, t, h, w = x.shape
    c = nc // up_time
    x = x.reshape(b, up_time, c, t, h, w).permute(0, 2, 3, 1, 4, 5).reshape(b, c, t * up_time, h, w)
    return x[:, :, 1:]


def _sdpa(q, k, v):
    """exact softmax attention, scale d**-0.5 (F.scaled_dot_product_attention / xformers default)."""
    s = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    return torch.matmul(torch.softmax(s, dim=-1), v)


# --------------------------------------------------------------------------------------------
# vae3d_sd3 family (CVVAESD3Model): models/vae_blocks3d_sd3.py + models/vae_models3d_sd3.py
# --------------------------------------------------------------------------------------------
def sd3_conv3d(x, sd: SD, pre: str, causal: bool, stride=1):
    """CausalConv3d (vae_blocks3d_sd3.py:81-104): replicate pad W(1,1) H(1,1) T(2,0) then conv pad 0.
    Conv3d (vae_blocks3d_sd3.py:16-46): nn.Conv3d(padding=1, padding_mode='replicate')."""
    pad = (1, 1, 1, 1, 2, 0) if causal else (1, 1, 1, 1, 1, 1)
    x = F.pad(x, pad, mode="replicate")
    return F.conv3d(x, sd[pre + ".weight"], sd[pre + ".bias"], stride=stride)


def sd3_resnet(x, sd: SD, pre: str, causal: bool):
    """ResnetBlock3D.forward, vae_blocks3d_sd3.py:517-569 (temb None, dropout 0, scale factor 1.0)."""
    h = _swish(_gn(x, sd, pre + ".norm1", 1e-6))
    h = sd3_conv3d(h, sd, pre + ".conv1", causal)
    h = _swish(_gn(h, sd, pre + ".norm2", 1e-6))
    h = _conv2d_extra_dim(h, sd[pre + ".conv2.weight"], sd[pre + ".conv2.bias"], padding=1)
    if (pre + ".conv_shortcut.weight") in sd:  # 1x1 Conv2dWithExtraDim, :500-515,564-565
        x = _conv2d_extra_dim(x, sd[pre + ".conv_shortcut.weight"], sd[pre + ".conv_shortcut.bias"], padding=0)
    return (x + h) / 1.0


def sd3_attention(x, sd: SD, pre: str):
    """AttentionWithExtraDim (vae_blocks3d_sd3.py:119-147) over diffusers Attention:
    per frame GN(32, eps 1e-6) over (C/32, H*W); to_q/k/v Linear; single-head SDPA; to_out.0; + residual."""
    b, c, t, h, w = x.shape
    f = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h * w)
    n = F.group_norm(f, 32, sd[pre + ".group_norm.weight"], sd[pre + ".group_norm.bias"], 1e-6)
    tok = n.transpose(1, 2)  # [bt, hw, c]
    q = F.linear(tok, sd[pre + ".to_q.weight"], sd[pre + ".to_q.bias"])
    k = F.linear(tok, sd[pre + ".to_k.weight"], sd[pre + ".to_k.bias"])
    v = F.linear(tok, sd[pre + ".to_v.weight"], sd[pre + ".to_v.bias"])
    o = _sdpa(q, k, v)
    o = F.linear(o, sd[pre + ".to_out.0.weight"], sd[pre + ".to_out.0.bias"])
    o = o.transpose(1, 2) + f
    return o.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)


def sd3_mid(x, sd: SD, pre: str, causal: bool, add_attention: bool = True):
    """UNetMidBlock3D.forward, vae_blocks3d_sd3.py:847-856."""
    x = sd3_resnet(x, sd, pre + ".resnets.0", causal)
    if add_attention:
        x = sd3_attention(x, sd, pre + ".attentions.0")
    return sd3_resnet(x, sd, pre + ".resnets.1", causal)


def sd3_encoder(x, sd: SD, cfg: dict, pre: str = "encoder"):
    """Encoder3D.forward, vae_models3d_sd3.py:162-208 (ctor :79-160)."""
    causal = cfg.get("causal_encoder", True)
    boc = cfg.get("block_out_channels", [128, 256, 512, 512])
    lpb = cfg.get("layers_per_block", 2)
    x = sd3_conv3d(x, sd, pre + ".conv_in", causal)
    for i in range(len(boc)):
        final = i == len(boc) - 1
        for j in range(lpb):
            x = sd3_resnet(x, sd, f"{pre}.down_blocks.{i}.resnets.{j}", causal)
        if not final:  # Downsample3D, vae_blocks3d_sd3.py:224-239; stride :198; down_time vae_models3d_sd3.py:115
            stride = 2 if (i % 2 == 0) else (1, 2, 2)
            x = sd3_conv3d(x, sd, f"{pre}.down_blocks.{i}.downsamplers.0.conv", causal, stride=stride)
    x = sd3_mid(x, sd, pre + ".mid_block", causal, cfg.get("mid_block_add_attention", True))
    x = _swish(_gn(x, sd, pre + ".conv_norm_out", 1e-6))
    return sd3_conv3d(x, sd, pre + ".conv_out", causal)


def sd3_upsample(x, sd: SD, pre: str, causal: bool, up_time: int):
    """Upsample3D.forward, vae_blocks3d_sd3.py:314-364: nearest x(1,2,2), conv to C*up_time, shuffle, drop."""
    x = F.interpolate(x, scale_factor=(1.0, 2.0, 2.0), mode="nearest")
    x = sd3_conv3d(x, sd, pre + ".conv", causal)
    return _time_shuffle(x, up_time)


def sd3_decoder(z, sd: SD, cfg: dict, pre: str = "decoder"):
    """Decoder3D.forward, vae_models3d_sd3.py:323-388 (ctor :234-321)."""
    causal = cfg.get("causal_decoder", False)
    boc = cfg.get("block_out_channels", [128, 256, 512, 512])
    lpb = cfg.get("layers_per_block", 2)
    x = sd3_conv3d(z, sd, pre + ".conv_in", causal)
    x = sd3_mid(x, sd, pre + ".mid_block", causal, cfg.get("mid_block_add_attention", True))
    for i in range(len(boc)):
        final = i == len(boc) - 1
        for j in range(lpb + 1):
            x = sd3_resnet(x, sd, f"{pre}.up_blocks.{i}.resnets.{j}", causal)
        if not final:  # up_time: vae_models3d_sd3.py:289
            x = sd3_upsample(x, sd, f"{pre}.up_blocks.{i}.upsamplers.0", causal, 2 if i % 2 == 0 else 1)
    x = _swish(_gn(x, sd, pre + ".conv_norm_out", 1e-6))
    return sd3_conv3d(x, sd, pre + ".conv_out", causal)


# --------------------------------------------------------------------------------------------
# vae3d family (CVVAEModel): models/vae_models.py
# --------------------------------------------------------------------------------------------
def v3_conv(x, sd: SD, pre: str, causal: bool, p: int):
    """causal: CausalConv3d vae_models.py:298-328 -- zero pad W,H by p, replicate pad T front 2p, conv pad 0.
    non-causal: plain nn.Conv3d(padding=p), zero pad on all faces (vae_models.py:361-362,952-955)."""
    w, b = sd[pre + ".weight"], sd[pre + ".bias"]
    if causal:
        x = F.pad(x, (p, p, p, p, 0, 0))
        if p > 0:
            x = F.pad(x, (0, 0, 0, 0, 2 * p, 0), mode="replicate")
        return F.conv3d(x, w, b)
    return F.conv3d(x, w, b, padding=p)


def v3_resnet(x, sd: SD, pre: str, causal: bool):
    """ResnetBlock3D.forward, vae_models.py:390-410 (eps 1e-5, half_3d conv2, nin_shortcut 1x1x1)."""
    h = _swish(_gn(x, sd, pre + ".norm1", 1e-5))
    h = v3_conv(h, sd, pre + ".conv1", causal, 1)
    h = _swish(_gn(h, sd, pre + ".norm2", 1e-5))
    h = _conv2d_extra_dim(h, sd[pre + ".conv2.weight"], sd[pre + ".conv2.bias"], padding=1)
    if (pre + ".nin_shortcut.weight") in sd:
        x = v3_conv(x, sd, pre + ".nin_shortcut", causal, 0)
    return x + h


def v3_downsample(x, sd: SD, pre: str, down_time: bool):
    """Downsample3D.forward, vae_models.py:251-263: zero pad W(0,1) H(0,1); replicate T(2,0); conv stride."""
    x = F.pad(x, (0, 1, 0, 1, 0, 0))
    x = F.pad(x, (0, 0, 0, 0, 2, 0), mode="replicate")
    return F.conv3d(x, sd[pre + ".conv.weight"], sd[pre + ".conv.bias"], stride=2 if down_time else (1, 2, 2))


def v3_upsample(x, sd: SD, pre: str, up_time: int):
    """Upsample3D.forward, vae_models.py:214-235 (decoder builds it non-causal, :936): nearest, zero pad
    W,H (1,1), replicate pad T (1,1), conv pad 0, shuffle + drop."""
    x = F.interpolate(x, scale_factor=(1.0, 2.0, 2.0), mode="nearest")
    x = F.pad(x, (1, 1, 1, 1, 0, 0))
    x = F.pad(x, (0, 0, 0, 0, 1, 1), mode="replicate")
    x = F.conv3d(x, sd[pre + ".conv.weight"], sd[pre + ".conv.bias"])
    return _time_shuffle(x, up_time)


def _v3_spatial_attn(x, sd: SD, pre: str):
    """MemoryEfficientAttnBlock.attention + proj_out (vae_models.py:500-537): per frame GN(eps 1e-5),
    1x1 Conv2d q/k/v, single-head attention scale C**-0.5, proj_out.  Returns frames [bt, c, h, w]."""
    b, c, t, h, w = x.shape
    f = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    n = F.group_norm(f, 32, sd[pre + ".norm.weight"], sd[pre + ".norm.bias"], 1e-5)
    q = F.conv2d(n, sd[pre + ".q.weight"], sd[pre + ".q.bias"]).flatten(2).transpose(1, 2)
    k = F.conv2d(n, sd[pre + ".k.weight"], sd[pre + ".k.bias"]).flatten(2).transpose(1, 2)
    v = F.conv2d(n, sd[pre + ".v.weight"], sd[pre + ".v.bias"]).flatten(2).transpose(1, 2)
    o = _sdpa(q, k, v).transpose(1, 2).reshape(b * t, c, h, w)
    return F.conv2d(o, sd[pre + ".proj_out.weight"], sd[pre + ".proj_out.bias"])


def v3_attn_spatial(x, sd: SD, pre: str):
    """MemoryEfficientAttnBlock.forward, vae_models.py:530-537 (encoder mid; == AttnBlock :463-470)."""
    b, c, t, h, w = x.shape
    o = _v3_spatial_attn(x, sd, pre)
    return x + o.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)


def v3_attn_spatial_temporal(x, sd: SD, pre: str):
    """MemoryEfficientAttnVideoBlock.forward, vae_models.py:619-629: spatial attention WITHOUT residual,
    then per pixel over T: LayerNorm(C) -> q_t/k_t/v_t -> attention -> proj_out_t (:573-587); one residual."""
    b, c, t, h, w = x.shape
    o = _v3_spatial_attn(x, sd, pre)  # [bt, c, h, w]
    s = o.reshape(b, t, c, h, w).permute(0, 3, 4, 1, 2).reshape(b * h * w, t, c)
    s = F.layer_norm(s, (c,), sd[pre + ".norm_t.weight"], sd[pre + ".norm_t.bias"], 1e-5)
    q = F.linear(s, sd[pre + ".q_t.weight"], sd[pre + ".q_t.bias"])
    k = F.linear(s, sd[pre + ".k_t.weight"], sd[pre + ".k_t.bias"])
    v = F.linear(s, sd[pre + ".v_t.weight"], sd[pre + ".v_t.bias"])
    o = F.linear(_sdpa(q, k, v), sd[pre + ".proj_out_t.weight"], sd[pre + ".proj_out_t.bias"])
    return x + o.reshape(b, h, w, t, c).permute(0, 4, 3, 1, 2)


def v3_encoder(x, sd: SD, cfg: dict, pre: str = "encoder"):
    """Encoder.forward, vae_models.py:790-823 (ctor :679-788); down_time = i_level % 2 == 0 (:750-754)."""
    causal = cfg.get("causal_encoder", True)
    ch_mult = cfg.get("ch_mult", [1, 2, 4, 4])
    nrb = cfg.get("num_res_blocks", 2)
    x = v3_conv(x, sd, pre + ".conv_in", causal, 1)
    for lvl in range(len(ch_mult)):
        for j in range(nrb):
            x = v3_resnet(x, sd, f"{pre}.down.{lvl}.block.{j}", causal)
        if lvl != len(ch_mult) - 1:
            x = v3_downsample(x, sd, f"{pre}.down.{lvl}.downsample", lvl % 2 == 0)
    x = v3_resnet(x, sd, pre + ".mid.block_1", causal)
    x = v3_attn_spatial(x, sd, pre + ".mid.attn_1")
    x = v3_resnet(x, sd, pre + ".mid.block_2", causal)
    x = _swish(_gn(x, sd, pre + ".norm_out", 1e-5))
    return v3_conv(x, sd, pre + ".conv_out", causal, 1)


def v3_decoder(z, sd: SD, cfg: dict, pre: str = "decoder"):
    """Decoder.forward, vae_models.py:960-1002 (ctor :826-944); up_time = i_level % 2 == 1 (:931-936)."""
    causal = cfg.get("causal_decoder", False)
    ch_mult = cfg.get("ch_mult", [1, 2, 4, 4])
    nrb = cfg.get("num_res_blocks", 2)
    x = v3_conv(z, sd, pre + ".conv_in", causal, 1)
    x = v3_resnet(x, sd, pre + ".mid.block_1", causal)
    x = v3_attn_spatial_temporal(x, sd, pre + ".mid.attn_1")
    x = v3_resnet(x, sd, pre + ".mid.block_2", causal)
    for lvl in reversed(range(len(ch_mult))):
        for j in range(nrb + 1):
            x = v3_resnet(x, sd, f"{pre}.up.{lvl}.block.{j}", causal)
        if lvl != 0:
            x = v3_upsample(x, sd, f"{pre}.up.{lvl}.upsample", 2 if lvl % 2 == 1 else 1)
    x = _swish(_gn(x, sd, pre + ".norm_out", 1e-5))
    return v3_conv(x, sd, pre + ".conv_out", causal, 1)


# --------------------------------------------------------------------------------------------
# wrapper level: temporal windows, spatial tiles, blending  (models/modeling_vae.py)
# --------------------------------------------------------------------------------------------
def blend_v(a, b, o):
    """modeling_vae.py:332-341 / 658-667: in place on b, fp32 ramp arange(o)/o (Appendix A.3)."""
    wgt = (torch.arange(o).view(1, 1, 1, -1, 1) / o).to(b.device)
    b[:, :, :, :o, :] = (1 - wgt) * a[:, :, :, -o:, :] + wgt * b[:, :, :, :o, :]
    return b


def blend_h(a, b, o):
    """modeling_vae.py:321-330 / 647-656."""
    wgt = (torch.arange(o).view(1, 1, 1, 1, -1) / o).to(b.device)
    b[:, :, :, :, :o] = (1 - wgt) * a[:, :, :, :, -o:] + wgt * b[:, :, :, :, :o]
    return b


def _spatial_tiled(x, net, tile: Optional[int], stride: int, overlap_out: int, stride_out: int):
    """spatial_tiled_encode / spatial_tiled_decode, modeling_vae.py:144-191, 230-277 (sd3 twin :470-603)."""
    if tile is None:
        return net(x)
    rows = []
    for i in range(0, x.shape[3], stride):
        cols = []
        for j in range(0, x.shape[4], stride):
            cols.append(net(x[:, :, :, i:i + tile, j:j + tile]))
            if j + tile >= x.shape[4]:
                break
        rows.append(cols)
        if i + tile >= x.shape[3]:
            break
    res = []
    for i, cols in enumerate(rows):
        rc = []
        for j, t in enumerate(cols):
            if i > 0:
                t = blend_v(rows[i - 1][j], t, overlap_out)
            if j > 0:
                t = blend_h(cols[j - 1], t, overlap_out)
            rc.append(t)
        res.append(rc)
    out_rows = []
    for i, cols in enumerate(res):
        for j, t in enumerate(cols):
            if i < len(res) - 1:
                t = t[:, :, :, :stride_out, :]
            if j < len(cols) - 1:
                t = t[:, :, :, :, :stride_out]
            cols[j] = t
        out_rows.append(torch.cat(cols, dim=4))
    return torch.cat(out_rows, dim=3)


def _windowed(x, fn, stride: Optional[int]):
    """tiled_encode / tiled_decode, modeling_vae.py:193-210, 279-296: windows [n*s, (n+1)*s], drop output
    frame 0 of windows n>0 (Appendix A.4: T=1 -> one round)."""
    if stride is None:
        return fn(x)
    n_rounds = math.ceil((x.shape[2] - 1) / stride) or 1
    outs = []
    for n in range(n_rounds):
        o = fn(x[:, :, n * stride:(n + 1) * stride + 1])
        outs.append(o if n == 0 else o[:, :, 1:])
    return torch.cat(outs, dim=2)


def _wrapper_consts(cfg: dict):
    """modeling_vae.py:84-109 / 410-435."""
    n = cfg.get("en_de_n_frames_a_time", 16)
    tnc = cfg.get("time_n_compress", 4)
    ts = cfg.get("tile_spatial_size", 576)
    snc = cfg.get("spatial_n_compress", 8)
    ratio = cfg.get("tile_overlap_ratio", 0.2222)
    enc_n = n if n is not None else None
    dec_n = n // tnc if n is not None else None
    px = ts if ts is not None else None
    lt = ts // snc if ts is not None else None
    return enc_n, dec_n, px, lt, ratio


def encode_moments(x, sd: SD, cfg: dict, family: str):
    """CVVAE*Model.encode up to the moments tensor, modeling_vae.py:212-228 / 538-554."""
    enc = (lambda t: sd3_encoder(t, sd, cfg)) if family == "sd3" else (lambda t: v3_encoder(t, sd, cfg))
    enc_n, _, px, lt, ratio = _wrapper_consts(cfg)
    if x.dim() == 4:
        nvf = cfg.get("num_video_frames")
        x = x.reshape(-1, nvf, *x.shape[1:]).permute(0, 2, 1, 3, 4) if nvf else x.unsqueeze(2)
    if px is None:
        sp = enc
    else:
        stride = round(px * (1 - ratio))
        ov = round(lt * ratio)
        sp = lambda t: _spatial_tiled(t, enc, px, stride, ov, lt - ov)  # noqa: E731
    return _windowed(x, sp, enc_n)


def decode_sample(z, sd: SD, cfg: dict, family: str, num_frames: Optional[int] = None):
    """CVVAE*Model.decode, modeling_vae.py:298-319 / 624-645."""
    dec = (lambda t: sd3_decoder(t, sd, cfg)) if family == "sd3" else (lambda t: v3_decoder(t, sd, cfg))
    _, dec_n, px, lt, ratio = _wrapper_consts(cfg)
    if z.dim() == 4:
        nvf = cfg.get("num_video_frames")
        tnc = cfg.get("time_n_compress", 4)
        nlf = (1 + (nvf - 1) // tnc) if nvf else None
        t = num_frames or nlf
        z = z.reshape(-1, t, *z.shape[1:]).permute(0, 2, 1, 3, 4) if t else z.unsqueeze(2)
    if lt is None:
        sp = dec
    else:
        stride = round(lt * (1 - ratio))
        ov = round(px * ratio)
        sp = lambda t: _spatial_tiled(t, dec, lt, stride, ov, px - ov)  # noqa: E731
    x = _windowed(z, sp, dec_n)
    if cfg.get("reshape_x_dim_to_4", False):
        x = x.permute(0, 2, 1, 3, 4).reshape(-1, x.shape[1], *x.shape[3:])
    return x


def posterior_mode(moments):
    """DiagonalGaussianDistribution.mode (lvdm/modules/distributions/distributions.py:72-73)."""
    return torch.chunk(moments, 2, dim=1)[0]


# --------------------------------------------------------------------------------------------------------
# the frozen 2-D "constraint" decoder of the training path (the SD3 image VAE decoder applied per frame
# to the 3-D VAE's latents: lvdm/models/autoencoder.py:1057-1069, configs/cvvae_sd3_constraint_training.yaml:40-51)
# --------------------------------------------------------------------------------------------------------
def c2d_resnet(x, sd: SD, pre: str):
    """ResnetBlock2D.forward, lvdm/modules/diffusionmodules/vae_blocks_sd3.py:368-421 (temb None, eps 1e-6, scale factor 1)."""
    h = _swish(_gn(x, sd, pre + ".norm1", 1e-6))
    h = F.conv2d(h, sd[pre + ".conv1.weight"], sd[pre + ".conv1.bias"], padding=1)
    h = _swish(_gn(h, sd, pre + ".norm2", 1e-6))
    h = F.conv2d(h, sd[pre + ".conv2.weight"], sd[pre + ".conv2.bias"], padding=1)
    if (pre + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[pre + ".conv_shortcut.weight"], sd[pre + ".conv_shortcut.bias"])
    return (x + h) / 1.0


def constraint_decoder(z, sd: SD, cfg: dict):
    """DecoderWith3DWrapper.forward (vae_models_sd3.py:390-398) over Decoder.forward (:297-362, eval path): 5-D latents
    [b,c,t,h,w] are decoded frame by frame ("b c t h w -> (b t) c h w"), 4-D ones directly."""
    five = z.dim() == 5
    if five:
        b, c, t, hh, ww = z.shape
        z = z.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww)
    boc = list(cfg.get("block_out_channels", [128, 256, 512, 512]))
    lpb = cfg.get("layers_per_block", 2)
    x = F.conv2d(z, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    x = c2d_resnet(x, sd, "mid_block.resnets.0")  # UNetMidBlock2D.forward, vae_blocks_sd3.py:669-681
    if cfg.get("mid_block_add_attention", True):
        x = sd3_attention(x.unsqueeze(2), sd, "mid_block.attentions.0").squeeze(2)  # same diffusers Attention, frames = batch
    x = c2d_resnet(x, sd, "mid_block.resnets.1")
    for i in range(len(boc)):  # UpDecoderBlock2D.forward, vae_blocks_sd3.py:536-547
        for j in range(lpb + 1):
            x = c2d_resnet(x, sd, f"up_blocks.{i}.resnets.{j}")
        if i != len(boc) - 1:  # Upsample2D.forward, vae_blocks_sd3.py:178-230: nearest x2, conv 3x3 zero pad
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            pre = f"up_blocks.{i}.upsamplers.0.conv"
            x = F.conv2d(x, sd[pre + ".weight"], sd[pre + ".bias"], padding=1)
    x = _swish(_gn(x, sd, "conv_norm_out", 1e-6))
    x = F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    if five:
        x = x.reshape(b, t, *x.shape[1:]).permute(0, 2, 1, 3, 4)
    return x

