"""Round 6 on the device:
  * the PLANAR fast-fp32 conv instances (conv_kernel.h Geo::PL: hi planes and bf6 code planes in separate LDS regions, eight fragments
    per wave): same products in the same order as the 256- / 128-pixel tiles of rounds 3-5 -> the stored outputs must agree BIT FOR
    BIT on every padding flavour, time fold, tile overhang and epilogue; the GroupNorm records (laid out per tile) must finalize to
    the same tables; and against fp32 F.conv3d (reference ops: models/vae_blocks3d_sd3.py:16-116, 517-569)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REP, ZERO = 1, 0
PC = ((2, 0), (1, 1), (1, 1))
P1 = ((1, 1), (1, 1), (1, 1))
P2D = ((0, 0), (1, 1), (1, 1))
# name, Cin, Cout, k, pad, mode_t, mode_hw, (B,T,H,W), planar tile ("TTxTHxTW:WMxWNxKG:KSUB"), 4-fragment tile, extras
PL_CASES = [
    ("enc128_causal_tfolds", 128, 128, (3, 3, 3), PC, REP, REP, (1, 6, 32, 64), "2x8x32:2x4x1:1", "2x4x32:2x4x1:1", dict(tfolds=True, stats=True)),
    ("enc128_odd_frames", 128, 128, (3, 3, 3), PC, REP, REP, (1, 5, 16, 64), "2x8x32:2x4x1:1", "2x4x32:2x4x1:1", dict(tfolds=True, stats=True)),
    ("dec256to128_sym", 256, 128, (3, 3, 3), P1, REP, REP, (1, 4, 24, 40), "2x8x32:2x4x1:1", "2x4x32:2x4x1:1", dict(tfolds=True)),
    ("vae3d_zero_time_pad", 128, 128, (3, 3, 3), P1, ZERO, ZERO, (2, 4, 16, 32), "2x8x32:2x4x1:1", "2x4x32:2x4x1:1", dict()),
    ("enc256_causal", 256, 256, (3, 3, 3), PC, REP, REP, (1, 3, 16, 64), "1x8x32:1x8x1:1", "1x4x32:1x8x1:1", dict(tfolds=True, stats=True)),
    ("mid512_overhang", 512, 512, (3, 3, 3), P1, REP, REP, (1, 2, 12, 40), "1x8x32:1x8x1:1", "1x4x32:1x8x1:1", dict()),
    ("c2d128_res_stats", 128, 128, (1, 3, 3), P2D, ZERO, ZERO, (1, 3, 32, 64), "1x16x32:2x4x1:2", "1x8x32:2x4x1:2", dict(res=True, stats=True)),
    ("c2d128_overhang", 128, 128, (1, 3, 3), P2D, ZERO, ZERO, (2, 2, 24, 40), "1x16x32:2x4x1:2", "1x8x32:2x4x1:2", dict(res=True)),
]


@pytest.mark.parametrize("case", PL_CASES, ids=[c[0] for c in PL_CASES])
def test_planar_fast_fp32_instances_reproduce_the_four_fragment_tiles(case, monkeypatch):
    import torch.nn.functional as F

    from cvvae_amd import _lib as L
    from cvvae_amd import ops
    from tests.test_gpu_grad3d import _pad3
    name, cin, cout, k, pad, mt, mhw, (B, T, H, W), tile_pl, tile_old, ex = case
    torch.manual_seed(len(name))
    x = torch.randn(B, T, H, W, cin).cuda() * 1.5 + 0.3
    w = torch.randn(cout, cin, *k) / (cin * k[0] * k[1] * k[2]) ** 0.5
    b = torch.randn(cout) * 0.1
    if ex.get("tfolds"):
        pw = ops.pack_weight_tfolds(w.cuda(), b.cuda(), fast="fp6")
    else:
        pw = ops.pack_weight(w.reshape(cout, cin, -1).cuda(), b.cuda(), k, fast="fp6")
    pw.act_bound = 8.0
    gam, bet = (1.0 + 0.2 * torch.randn(cin)).cuda(), (0.1 * torch.randn(cin)).cuda()
    gn = ops.gn_stats(x, gam, bet, 1e-6)
    kw = dict(pad=pad, pad_mode_t=mt, pad_mode_hw=mhw, prologue=L.PRO_GN_SILU, gn=gn)
    if ex.get("res"):
        kw["residual"] = torch.randn(B, T, H, W, cout).cuda()
    if ex.get("stats"):
        kw["gn_out"] = 32
    one, zero = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    res, names = {}, {}
    for label, tile in (("planar", tile_pl), ("four", tile_old)):
        monkeypatch.setenv("CVVAE_CONV_FORCE", tile)
        seen = []
        ops.PROFILE = lambda d, pw_, launch: (seen.append(ops.conv_kernel_name(d)), launch())
        try:
            out = ops.conv(x, pw, **kw)
        finally:
            ops.PROFILE = None
        names[label] = seen
        if isinstance(out, tuple):
            res[label] = (out[0], ops.gn_finalize(out[1], one, zero, 1e-6))
        else:
            res[label] = (out, None)
    monkeypatch.delenv("CVVAE_CONV_FORCE")
    tag = "_t" + tile_pl.split(":")[0] + "_"
    assert names["planar"] and tag in names["planar"][0] and names["planar"][0].endswith("_xq6"), names
    assert names["four"] and tag not in names["four"][0] and names["four"][0].endswith("_xq6"), names
    ya, yb = res["planar"][0], res["four"][0]
    assert torch.equal(ya, yb), (name, names, float((ya - yb).abs().max()))
    if ex.get("stats"):  # other tiles -> other records, merged in another order: the tables agree to fp32 rounding
        for ta, tb in zip(res["planar"][1], res["four"][1]):
            assert torch.allclose(ta, tb, rtol=2e-5, atol=2e-6), (name, float((ta - tb).abs().max()))
    # against fp32 conv3d over the fp32 GroupNorm + SiLU of the same input (the fast rung: ~2^-13 relative per product)
    a = F.silu(x * gn[0].view(B, 1, 1, 1, cin) + gn[1].view(B, 1, 1, 1, cin)).cpu()
    ref = F.conv3d(_pad3(a.permute(0, 4, 1, 2, 3), pad, mt, mhw), w, b)
    if ex.get("res"):
        ref = ref + kw["residual"].cpu().permute(0, 4, 1, 2, 3)
    got = ya.cpu().permute(0, 4, 1, 2, 3)
    assert float((got - ref).abs().max()) <= 4e-4 * float(ref.abs().max()), float((got - ref).abs().max())
