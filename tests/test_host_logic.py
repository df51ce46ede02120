"""Host logic of the drop-in wrapper (temporal windows, spatial tiles, in-place blending, 4-D<->5-D reshapes) on CPU.
The HIP engine cannot run here (and has no CPU fallback), so the wrapper's `encoder` / `decoder` / `_blend` are
replaced by TEST DOUBLES defined in this file; the expected values come from the oracle's restatement of
modeling_vae.py driven by the same doubles, and -- when /root/reference is mounted -- from the reference wrapper."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import cvvae_oracle as O
from oracle.ref_loader import load_reference, reference_available


class StubEnc(torch.nn.Module):
    """[B,3,T,H,W] -> [B,8,1+(T-1)//4,H/8,W/8]; position dependent so that tile/window mix-ups are visible."""

    def forward(self, x):
        t = x.shape[2]
        tt = 1 + (t - 1) // 4
        idx = [min(4 * i, t - 1) for i in range(tt)]
        y = F.avg_pool3d(x[:, :, idx], (1, 8, 8))
        y = torch.cat([y, y.flip(1), y[:, :2] * 0.5], dim=1)
        return y + 0.01 * y.shape[-1] + 0.001 * y.shape[-2]


class StubDec(torch.nn.Module):
    """[B,4,T',h,w] -> [B,3,1+4(T'-1),8h,8w]"""

    def forward(self, z, **kw):
        tt = z.shape[2]
        idx = [i // 4 if i else 0 for i in range(1 + 4 * (tt - 1))]
        idx = [min((i + 3) // 4, tt - 1) for i in range(1 + 4 * (tt - 1))]
        y = F.interpolate(z[:, :3, idx], scale_factor=(1, 8, 8), mode="nearest")
        return y * 1.5 + 0.01 * z.shape[-1]


def torch_blend(a, b, o, axis):
    return O.blend_v(a, b, o) if axis == 0 else O.blend_h(a, b, o)


def make(family, **cfg):
    import cvvae_amd
    cls = cvvae_amd.CVVAESD3Model if family == "sd3" else cvvae_amd.CVVAEModel
    m = cls(**cfg)
    m.encoder, m.decoder = StubEnc(), StubDec()
    m._blend = staticmethod(torch_blend)
    return m


def expected_encode(x, cfg):
    enc_n, _, px, lt, ratio = O._wrapper_consts(cfg)
    enc = StubEnc()
    if px is None:
        sp = enc
    else:
        ov = round(lt * ratio)
        sp = lambda t: O._spatial_tiled(t, enc, px, round(px * (1 - ratio)), ov, lt - ov)  # noqa: E731
    return O._windowed(x, sp, enc_n)


def expected_decode(z, cfg):
    _, dec_n, px, lt, ratio = O._wrapper_consts(cfg)
    dec = StubDec()
    if lt is None:
        sp = dec
    else:
        ov = round(px * ratio)
        sp = lambda t: O._spatial_tiled(t, dec, lt, round(lt * (1 - ratio)), ov, px - ov)  # noqa: E731
    return O._windowed(z, sp, dec_n)


CFGS = [
    {},                                                     # shipped defaults (576-px tiles, 16-frame windows)
    {"tile_spatial_size": 144},                             # small tiles -> 2-D tile grids on small inputs
    {"tile_spatial_size": None},
    {"en_de_n_frames_a_time": None},
    {"en_de_n_frames_a_time": 8, "tile_spatial_size": 64},  # inconsistent pixel/latent strides, as the maths gives them
]


@pytest.mark.parametrize("family", ["sd3", "vae3d"])
@pytest.mark.parametrize("cfg", CFGS)
@pytest.mark.parametrize("shape", [(1, 3, 1, 64, 64), (2, 3, 17, 160, 200), (1, 3, 33, 96, 320), (1, 3, 21, 296, 152)])
def test_encode_decode_tiling_matches_restatement(family, cfg, shape):
    m = make(family, **cfg)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(shape, generator=g) * 2 - 1
    moments = m.encode(x).latent_dist.parameters
    exp = expected_encode(x, cfg)
    assert moments.shape == exp.shape and torch.equal(moments, exp)
    z = moments[:, :4]
    y = m.decode(z).sample
    expd = expected_decode(z, cfg)
    assert y.shape == expd.shape and torch.equal(y, expd)
    assert y.shape[2] == shape[2]
    if cfg.get("tile_spatial_size", 576) != 64:  # (64-px tiles give inconsistent strides; only equality with the restatement matters)
        assert y.shape[3:] == x.shape[3:]


@pytest.mark.skipif(not reference_available(), reason="/root/reference not mounted")
@pytest.mark.parametrize("family", ["sd3", "vae3d"])
@pytest.mark.parametrize("cfg", CFGS[:3])
def test_tiling_matches_reference_wrapper(family, cfg):
    ref = load_reference()
    rcls = ref.CVVAESD3Model if family == "sd3" else ref.CVVAEModel
    with torch.device("meta"):
        r = rcls(**cfg)
    r.encoder, r.decoder = StubEnc(), StubDec()
    m = make(family, **cfg)
    g = torch.Generator().manual_seed(1)
    x = torch.rand((1, 3, 33, 200, 328), generator=g) * 2 - 1
    a = m.encode(x).latent_dist.parameters
    b = r.encode(x).latent_dist.parameters
    assert torch.equal(a, b)
    assert torch.equal(m.decode(a[:, :4]).sample, r.decode(b[:, :4]).sample)


def test_4d_inputs_and_tuple_returns():
    m = make("sd3", num_video_frames=5, reshape_x_dim_to_4=True, tile_spatial_size=None)
    x5 = torch.rand(2, 3, 5, 32, 32)
    x4 = x5.permute(0, 2, 1, 3, 4).reshape(10, 3, 32, 32)
    (post,) = m.encode(x4, return_dict=False)
    assert torch.equal(post.parameters, m.encode(x5).latent_dist.parameters)
    z5 = post.mode()[:, :4]
    z4 = z5.permute(0, 2, 1, 3, 4).reshape(-1, 4, 4, 4)
    (y,) = m.decode(z4, return_dict=False)            # num_latent_frames from num_video_frames (:634)
    assert y.shape == (10, 3, 32, 32)                  # reshape_x_dim_to_4 (:639)
    m2 = make("vae3d", tile_spatial_size=None)
    img = torch.rand(3, 3, 32, 32)                     # T=1 image mode: 'b c h w -> b c () h w' (:220)
    assert m2.encode(img).latent_dist.mean.shape == (3, 4, 1, 4, 4)
    assert m2.decode(torch.rand(6, 4, 4, 4), num_frames=2).sample.shape == (3, 3, 5, 32, 32)


def test_window_schedule():
    from cvvae_amd.modeling import _CVVAEBase
    assert _CVVAEBase._windows(1, 16) == [(0, 17)]
    assert _CVVAEBase._windows(17, 16) == [(0, 17)]
    assert _CVVAEBase._windows(33, 16) == [(0, 17), (16, 33)]
    assert _CVVAEBase._windows(129, 16) == [(16 * n, 16 * n + 17) for n in range(8)]
    assert _CVVAEBase._windows(9, 4) == [(0, 5), (4, 9)]
    # cfg 4 (720x1280): 2 x 3 spatial tiles per window -> 48 encoder calls for T=129
    calls = []

    class Count(torch.nn.Module):
        def forward(self, t):
            calls.append(tuple(t.shape[2:]))
            return torch.zeros(t.shape[0], 8, 1 + (t.shape[2] - 1) // 4, math.ceil(t.shape[3] / 8), math.ceil(t.shape[4] / 8))

    m = make("sd3")
    m.encoder = Count()
    m.encode(torch.zeros(1, 3, 129, 720, 1280))
    assert len(calls) == 48
    assert set(c[1:] for c in calls) == {(576, 576), (576, 384), (272, 576), (272, 384)}


def test_posterior():
    from cvvae_amd.modeling import DiagonalGaussianDistribution
    p = torch.randn(2, 8, 3, 4, 4) * 20
    d = DiagonalGaussianDistribution(p)
    assert torch.equal(d.mode(), p[:, :4])
    assert d.logvar.max() <= 20 and d.logvar.min() >= -30
    g1, g2 = torch.Generator().manual_seed(7), torch.Generator().manual_seed(7)
    assert torch.equal(d.sample(generator=g1), d.mean + d.std * torch.randn(d.mean.shape, generator=g2))


@pytest.mark.parametrize("shape", [(40, 64, 24, 40), (37, 53, 64, 96), (48, 80, 48, 50), (30, 30, 7, 11), (20, 36, 33, 20), (270, 480, 144, 256)])
def test_resize_tables_reproduce_torch_uint8_antialiased_bilinear(shape):
    """ops.resize_tables (the host side of cvvae_resize_u8_axis: filter support, int16 weight scale, rounding) applied in plain integer
    arithmetic -- width pass, then height pass, as the kernel does -- equals F.interpolate(uint8, mode='bilinear', antialias=True),
    i.e. what the scripts' transforms.Resize computes on the uint8 clip (cvvae_inference_video.py:14-16,28), bit for bit"""
    from cvvae_amd import ops
    H, W, oh, ow = shape
    x = torch.randint(0, 256, (2, 3, H, W), generator=torch.Generator().manual_seed(H * W), dtype=torch.uint8)
    ref = F.interpolate(x, size=(oh, ow), mode="bilinear", antialias=True)
    cur = x.long()
    for axis, (ni, no) in ((3, (W, ow)), (2, (H, oh))):
        if ni == no:
            continue
        xmin, xsize, wi, ks, prec = ops.resize_tables(ni, no)
        assert all(len(r) == ks for r in wi) and 1 <= prec <= 22
        cur = cur.movedim(axis, -1)
        out = torch.zeros(cur.shape[:-1] + (no,), dtype=torch.long)
        for i in range(no):
            acc = torch.full(cur.shape[:-1], 1 << (prec - 1), dtype=torch.long)
            for j in range(xsize[i]):
                acc += wi[i][j] * cur[..., xmin[i] + j]
            out[..., i] = (acc >> prec).clamp(0, 255)
        cur = out.movedim(-1, axis)
    assert torch.equal(cur, ref.long())
