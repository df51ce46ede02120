"""Round 6 on the device:
  * the PLANAR fast-fp32 conv instances (conv_kernel.h Geo::PL: hi planes and bf6 code planes in separate LDS regions, eight fragments
    per wave): same products in the same order as the 256- / 128-pixel tiles of rounds 3-5 -> the stored outputs must agree BIT FOR
    BIT on every padding flavour, time fold, tile overhang and epilogue; the GroupNorm records (laid out per tile) must finalize to
    the same tables; and against fp32 F.conv3d (reference ops: models/vae_blocks3d_sd3.py:16-116, 517-569);
  * the four-wave conv instances as a per-launch DESCRIPTOR field (ABI 13: the library keeps no state), and the bit-for-bit
    repetition check of their fused GroupNorm records under co-residency that used to run at model load;
  * the fp6-correction form of the folded upsample convs of a fast-fp32 model, with the operand's bound read from the device
    (cvvae_conv_desc.act_bound_dev; reference op: Upsample3D, models/vae_blocks3d_sd3.py:314-364)."""
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
    # the 2 x 4 register-block instances (two 32-channel blocks x four fragments per wave: "...:2" = N-blocks per wave)
    ("nb2_enc128_causal_tfolds", 128, 128, (3, 3, 3), PC, REP, REP, (1, 6, 32, 64), "2x8x32:4x2x1:1:2", "2x4x32:2x4x1:1", dict(tfolds=True, stats=True)),
    ("nb2_enc128_odd_frames", 128, 128, (3, 3, 3), PC, REP, REP, (1, 5, 16, 64), "2x8x32:4x2x1:1:2", "2x4x32:2x4x1:1", dict(tfolds=True, stats=True)),
    ("nb2_dec256to128_sym", 256, 128, (3, 3, 3), P1, REP, REP, (1, 4, 24, 40), "2x8x32:4x2x1:1:2", "2x4x32:2x4x1:1", dict(tfolds=True)),
    ("nb2_enc256_causal", 256, 256, (3, 3, 3), PC, REP, REP, (1, 3, 16, 64), "1x8x32:2x4x1:1:2", "1x4x32:1x8x1:1", dict(tfolds=True, stats=True)),
    ("nb2_mid512_overhang", 512, 512, (3, 3, 3), P1, ZERO, ZERO, (2, 2, 12, 40), "1x8x32:2x4x1:1:2", "1x4x32:1x8x1:1", dict()),
    ("nb2_c2d128_res_stats", 128, 128, (1, 3, 3), P2D, ZERO, ZERO, (1, 3, 32, 64), "1x16x32:4x2x1:2:2", "1x8x32:2x4x1:2", dict(res=True, stats=True)),
    ("nb2_c2d128_overhang", 128, 128, (1, 3, 3), P2D, ZERO, ZERO, (2, 2, 24, 40), "1x16x32:4x2x1:2:2", "1x8x32:2x4x1:2", dict(res=True)),
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
    tag = "_t" + tile_pl.split(":")[0] + "_w" + tile_pl.split(":")[1] + "_"
    suffix = "_xq6nb2" if tile_pl.count(":") == 3 else "_xq6"
    assert names["planar"] and tag in names["planar"][0] and names["planar"][0].endswith(suffix), names
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


def _c2d128_desc(four_wave, dtype=torch.bfloat16):
    from cvvae_amd import _lib as L
    from cvvae_amd import ops
    d = L.ConvDesc()
    d.dtype = ops._dt(dtype)
    d.B, d.Ti, d.Hi, d.Wi, d.Cin, d.in_pix_stride = 1, 17, 512, 512, 128, 128
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW = 1, 3, 3, 1, 1, 1
    d.pad_t, d.pad_h, d.pad_w, d.prologue, d.gn_rows_per_batch = 0, 1, 1, 1, 1
    d.To, d.Ho, d.Wo, d.Cout, d.out_pix_stride, d.alpha = 17, 512, 512, 128, 128, 1.0
    d.four_wave = four_wave
    return d


def test_four_wave_is_a_descriptor_field(monkeypatch):
    """ABI 13: `cvvae_conv_desc.four_wave` decides per LAUNCH whether the four-wave instances are candidates (until ABI 12 a
    process-wide switch inside the library, cvvae_conv_set_four_wave); the Python layer sets it unless CVVAE_FOUR_WAVE=0; a model
    pass gives the same latents either way to the last-bit rounding of the residual pre-accumulation"""
    import cvvae_amd
    from cvvae_amd import _lib as L
    from cvvae_amd import ops
    from oracle import parity as P
    from oracle.seeded import seeded_input
    lib = L.load()
    assert "cvvae_conv_set_four_wave" not in L.PROTOTYPES
    n0, n1 = ops.conv_kernel_name(_c2d128_desc(0)), ops.conv_kernel_name(_c2d128_desc(1))
    assert "w2x4x1" in n0 and "w1x4x1" in n1, (n0, n1)
    assert lib.cvvae_conv_gn_slabs(_c2d128_desc(0), 32) > 0 and lib.cvvae_conv_gn_slabs(_c2d128_desc(1), 32) > 0
    bad = _c2d128_desc(2)
    assert lib.cvvae_conv_gn_slabs(bad, 32) < 0 and ops.conv_kernel_name(bad) is None
    m = cvvae_amd.CVVAESD3Model()
    P.load_seeded(m, 0)
    m = m.to(torch.bfloat16).cuda().eval().requires_grad_(False)
    x = seeded_input((1, 3, 5, 128, 128), 3).to(torch.bfloat16).cuda()
    seen = {}
    for env in ("0", "1"):
        monkeypatch.setenv("CVVAE_FOUR_WAVE", env)
        names = []
        ops.PROFILE = lambda d, pw_, launch: (names.append(ops.conv_kernel_name(d)), launch())
        try:
            seen[env] = (m.encode(x).latent_dist.mode(), names)
        finally:
            ops.PROFILE = None
    assert not any("w1x4x1" in (n or "") for n in seen["0"][1])
    assert float((seen["0"][0].float() - seen["1"][0].float()).abs().max()) <= 0.05


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bfloat16", "float16"])
def test_four_wave_records_are_reproducible_under_co_residency(dtype):
    """The check that rounds 3-5 ran at model load (and tools/probes/nw4_stress.py runs at length), as a test: the per-frame
    128-channel launch with residual + fused statistics at 512x512 (every CU double-occupied for several rounds) repeated -- every
    GroupNorm record and every output bit-identical between the repetitions, the finalized tables equal to the 8-wave instance's.
    Round 2 once saw ~1 record in 10^4 differ on one box of the pool; no box since has."""
    from cvvae_amd import _lib as L
    from cvvae_amd import ops
    g = torch.Generator(device="cuda").manual_seed(4)
    shape = (1, 4, 512, 512, 128)
    x = torch.randn(shape, generator=g, device="cuda", dtype=torch.float32).to(dtype)
    res = torch.randn(shape, generator=g, device="cuda", dtype=torch.float32).to(dtype)
    gsc = (1.0 + 0.1 * torch.randn((1, 128), generator=g, device="cuda")).contiguous()
    gsh = (0.1 * torch.randn((1, 128), generator=g, device="cuda")).contiguous()
    w = (torch.randn((128, 128, 9), generator=g, device="cuda") / (128 * 9) ** 0.5).to(dtype)
    pw = ops.pack_weight(w, torch.zeros(128, device="cuda"), (1, 3, 3))
    kw = dict(pad=P2D, pad_mode_hw=ZERO, prologue=L.PRO_GN_SILU, gn=(gsc, gsh), residual=res, gn_out=32)
    one, zero = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    old = os.environ.get("CVVAE_FOUR_WAVE")
    names = []
    try:
        os.environ["CVVAE_FOUR_WAVE"] = "0"
        y8, p8 = ops.conv(x, pw, **kw)
        t8 = ops.gn_finalize(p8, one, zero, 1e-6)
        os.environ["CVVAE_FOUR_WAVE"] = "1"
        ops.PROFILE = lambda d, pw_, launch: (names.append(ops.conv_kernel_name(d)), launch())
        y4, p4 = ops.conv(x, pw, **kw)
        ops.PROFILE = None
        assert names and "w1x4x1" in names[-1], names
        bad = 0
        for _ in range(24):
            yr, pr = ops.conv(x, pw, **kw)
            bad += int(not torch.equal(pr.buf, p4.buf)) + int(not torch.equal(yr, y4))
        assert bad == 0, f"{bad} of 48 comparisons differ on {torch.cuda.get_device_name()}: irreproducible four-wave results -- please report the device"
        t4 = ops.gn_finalize(p4, one, zero, 1e-6)
        assert torch.allclose(t4[0], t8[0], rtol=2e-5) and torch.allclose(t4[1], t8[1], rtol=2e-5, atol=2e-6)
        assert float((y4.float() - y8.float()).abs().max()) <= (2.0 ** -4 if dtype == torch.bfloat16 else 2.0 ** -7)
    finally:
        ops.PROFILE = None
        if old is None:
            os.environ.pop("CVVAE_FOUR_WAVE", None)
        else:
            os.environ["CVVAE_FOUR_WAVE"] = old


@pytest.mark.parametrize("shuffle", [False, True])
@pytest.mark.parametrize("time_folds", [False, True])
def test_fp6_upsample_folded_with_a_device_side_bound(shuffle, time_folds):
    """the folded upsample conv in the fp6 form: no GroupNorm bounds its operand (the residual stream), so the kernel reads the bound
    from the device (one float, a max-abs reduction on the producer's stream) -- same tolerance as the bf8 form; a bound 4x too
    loose or 100x too tight degrades gracefully (saturation hits the CORRECTION terms only)"""
    from tests.test_gpu_ops import FAST_ULP, REP, _ops, run_conv_case
    L = _ops()[1]
    args = (torch.float32, 256, 512 if shuffle else 256, (3, 3, 3), (1, 1, 1), ((1, 1), (1, 1), (1, 1)), REP, REP, (1, 3, 12, 20))
    kw = dict(ups=2, out_mode=L.OUT_TIME_SHUFFLE if shuffle else L.OUT_NDHWC, time_folds=time_folds)
    e6 = run_conv_case(*args, tol=2 * FAST_ULP, fast="fp6", **kw)
    e8 = run_conv_case(*args, tol=2 * FAST_ULP, fast=True, **kw)
    loose = run_conv_case(*args, tol=1.0, fast="fp6", act_bound=4.0, **kw)
    tight = run_conv_case(*args, tol=1.0, fast="fp6", act_bound=0.01, **kw)
    f16 = run_conv_case(torch.float16, *args[1:], tol=1.0, **kw)
    print(f"\nfolded upsample shuffle={shuffle} tfolds={time_folds}: fp6 {e6:.2e} bf8 {e8:.2e} fp6 loose {loose:.2e} tight {tight:.2e} fp16 model {f16:.2e}")
    assert e6 <= 1.5 * e8 + 1e-7 and loose <= 2.5 * e8 + 1e-7 and tight <= 1.5 * f16


def test_fp6_upsample_needs_exactly_one_bound():
    from cvvae_amd import _lib as L
    from cvvae_amd import ops
    lib = L.load()
    w = torch.randn(128, 128, 3, 3, 3).cuda() / 60
    pw = ops.pack_weight_upfold(w, None, fast="fp6")
    assert pw.dt == L.F32Q6
    x = torch.randn(1, 2, 8, 32, 128).cuda()
    with pytest.raises(ValueError):
        ops.conv(x, pw, pad=P1, pad_mode_t=REP, pad_mode_hw=REP, upsample2x=2)
    bound = torch.linalg.vector_norm(x.reshape(-1), float("inf")).reshape(1)
    y = ops.conv(x, pw, pad=P1, pad_mode_t=REP, pad_mode_hw=REP, upsample2x=2, act_bound_dev=bound)
    assert tuple(y.shape) == (1, 2, 16, 64, 128) and bool(torch.isfinite(y).all())
    d = _c2d128_desc(0, torch.float32)
    d.dtype, d.act_bound, d.act_bound_dev = L.F32Q6, 8.0, bound.data_ptr()
    assert lib.cvvae_conv_gn_slabs(d, 32) < 0  # both bounds: an argument error


@pytest.mark.parametrize("shuffle", [False, True])
def test_register_block_upsample_instance_reproduces_the_four_fragment_tile(shuffle, monkeypatch):
    """the folded upsample conv (fp6, device-side bound) on the 2 x 4 register-block planar instance vs the tile of rounds 3-5:
    same products in the same order -> identical bits"""
    from cvvae_amd import _lib as L
    from cvvae_amd import ops
    torch.manual_seed(3)
    cout = 512 if shuffle else 256
    x = torch.randn(1, 3, 24, 40, 256).cuda()
    pw = ops.pack_weight_upfold((torch.randn(cout, 256, 3, 3, 3) / 83).cuda(), torch.randn(cout).cuda() * 0.1, time_folds=True, fast="fp6")
    bound = torch.linalg.vector_norm(x.reshape(-1), float("inf")).reshape(1)
    out, names = {}, {}
    # (the 16-channel-chunk tile of rounds 3-5: the 32-channel one walks (chunk, time group, k16, tap) in another order -- 3e-6 apart)
    for label, tile in (("nb2", "1x8x32:2x4x1:1:2"), ("four", "1x4x32:1x8x1:1")):
        monkeypatch.setenv("CVVAE_CONV_FORCE", tile)
        seen = []
        ops.PROFILE = lambda d, pw_, launch: (seen.append(ops.conv_kernel_name(d)), launch())
        try:
            y, part = ops.conv(x, pw, pad=P1, pad_mode_t=REP, pad_mode_hw=REP, upsample2x=2,
                               out_mode=L.OUT_TIME_SHUFFLE if shuffle else L.OUT_NDHWC, gn_out=32, act_bound_dev=bound)
        finally:
            ops.PROFILE = None
        out[label], names[label] = y, seen
    monkeypatch.delenv("CVVAE_CONV_FORCE")
    assert names["nb2"][0].endswith("_xq6nb2") and names["four"][0].endswith("_xq6") and "ups2" in names["nb2"][0], names
    assert torch.equal(out["nb2"], out["four"]), float((out["nb2"] - out["four"]).abs().max())


# ---------------------------------------------------------------------------------------------------------------------------------
# small-frame tiles (conv_table.h G14) and the starved-grid term of the instance cost model
# ---------------------------------------------------------------------------------------------------------------------------------
# name, Cin, Cout, k, (B,T,H,W), small tile, the tile of rounds 1-5, extras
SMALL_CASES = [
    ("c2d512_32_res_stats_128px", 512, 512, (1, 3, 3), (1, 1, 32, 32), "1x4x32:2x4x1:2", "1x8x32:2x4x1:2", dict(res=True, stats=True)),
    ("c2d512_32_res_stats_64px", 512, 512, (1, 3, 3), (1, 1, 32, 32), "1x2x32:2x4x1:2", "1x8x32:2x4x1:2", dict(res=True, stats=True)),
    ("c2d256_overhang_64px_all_n", 256, 256, (1, 3, 3), (2, 2, 20, 40), "1x2x32:1x8x1:2", "1x8x32:1x8x1:2", dict(stats=True)),
    ("c2d256to512_overhang", 256, 512, (1, 3, 3), (1, 3, 12, 72), "1x2x32:2x4x1:2", "1x8x32:2x4x1:2", dict()),
    ("c2d512_64ch_chunks", 512, 512, (1, 3, 3), (1, 1, 32, 32), "1x2x32:1x8x1:4", "1x8x32:1x8x1:4", dict(stats=True)),
    ("c2d128to8_conv_out_like", 128, 8, (1, 3, 3), (1, 1, 32, 32), "1x2x32:2x4x1:2", "1x8x32:2x4x1:2", dict()),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bfloat16", "float16"])
@pytest.mark.parametrize("case", SMALL_CASES, ids=[c[0] for c in SMALL_CASES])
def test_small_frame_tiles_reproduce_the_256_pixel_tile(case, dtype, monkeypatch):
    """64- and 128-pixel tiles of the per-frame 3x3 conv (image mode, the 32x32 / 64x64 levels of a short clip): every output is
    accumulated over the same chunks and taps in the same order as under the 256-pixel tile -> bit-equal outputs (residual pre-accumulated
    at the same chunk); the GroupNorm records are laid out per tile, their tables agree to fp32 rounding.
    (reference op: ResnetBlock2D conv2 + residual, lvdm/modules/diffusionmodules/model.py:82-143)"""
    from cvvae_amd import _lib as L
    from cvvae_amd import ops
    name, cin, cout, k, (B, T, H, W), tile_small, tile_old, ex = case
    torch.manual_seed(len(name))
    x = (torch.randn(B, T, H, W, cin) * 1.5 + 0.3).to(dtype).cuda()
    w = (torch.randn(cout, cin, *k) / (cin * 9) ** 0.5).to(dtype)
    b = torch.randn(cout) * 0.1
    pw = ops.pack_weight(w.reshape(cout, cin, -1).cuda(), b.cuda(), k)
    gam, bet = (1.0 + 0.2 * torch.randn(cin)).cuda(), (0.1 * torch.randn(cin)).cuda()
    gn = ops.gn_stats(x, gam, bet, 1e-6)
    kw = dict(pad=P2D, pad_mode_t=ZERO, pad_mode_hw=ZERO, prologue=L.PRO_GN_SILU, gn=gn)
    if ex.get("res"):
        kw["residual"] = torch.randn(B, T, H, W, cout).to(dtype).cuda()
    if ex.get("stats"):
        kw["gn_out"] = 32
    one, zero = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    res, names = {}, {}
    for label, tile in (("small", tile_small), ("old", tile_old)):
        monkeypatch.setenv("CVVAE_CONV_FORCE", tile)
        seen = []
        ops.PROFILE = lambda d, pw_, launch: (seen.append(ops.conv_kernel_name(d)), launch())
        try:
            out = ops.conv(x, pw, **kw)
        finally:
            ops.PROFILE = None
        names[label] = seen
        res[label] = (out[0], ops.gn_finalize(out[1], one, zero, 1e-6)) if isinstance(out, tuple) else (out, None)
    monkeypatch.delenv("CVVAE_CONV_FORCE")
    tag = "_t" + tile_small.split(":")[0] + "_w" + tile_small.split(":")[1] + "_"
    assert names["small"] and tag in names["small"][0], names
    assert names["old"] and "_t1x8x32_" in names["old"][0], names
    ya, yb = res["small"][0], res["old"][0]
    if ex.get("res"):
        # the residual joins a fragment's accumulator at one of the first K chunks -- which one depends on the fragments per wave
        # (conv_kernel.h res_pre): the same sum in another order, one rounding of the stored type apart at most
        ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        d = (ya.float() - yb.float()).abs()
        assert float((d / yb.float().abs().clamp_min(1.0)).max()) <= 2 * ulp, (name, float(d.max()))
        assert float((d > 0).float().mean()) < 0.25, float((d > 0).float().mean())
    else:
        assert torch.equal(ya, yb), (name, names, float((ya.float() - yb.float()).abs().max()))
    if ex.get("stats"):
        for ta, tb in zip(res["small"][1], res["old"][1]):
            assert torch.allclose(ta, tb, rtol=2e-5, atol=2e-6), (name, float((ta - tb).abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bfloat16", "float16"])
@pytest.mark.parametrize("tile", ["1x2x32:1x4x2:4", "1x4x32:1x4x2:4"], ids=["64px", "128px"])
def test_small_frame_k_group_tiles(tile, dtype, monkeypatch):
    """The K-group forms of the small tiles (two groups of waves take half of every 64-channel chunk each, accumulators exchanged
    through LDS): another summation order than the all-K tiles -- the stored outputs agree with the 256-pixel tile's to a rounding
    of the stored type, residual and fused GroupNorm statistics included, and with fp32 conv2d over the same 16-bit operands.
    (reference op: ResnetBlock2D conv2 + residual, lvdm/modules/diffusionmodules/model.py:82-143)"""
    import torch.nn.functional as F

    from cvvae_amd import _lib as L
    from cvvae_amd import ops
    cin, cout, (B, T, H, W) = 512, 256, (2, 1, 20, 40)
    torch.manual_seed(23)
    x = (torch.randn(B, T, H, W, cin) * 1.5 + 0.3).to(dtype).cuda()
    w = (torch.randn(cout, cin, 1, 3, 3) / (cin * 9) ** 0.5).to(dtype)
    b = torch.randn(cout) * 0.1
    pw = ops.pack_weight(w.reshape(cout, cin, -1).cuda(), b.cuda(), (1, 3, 3))
    gam, bet = (1.0 + 0.2 * torch.randn(cin)).cuda(), (0.1 * torch.randn(cin)).cuda()
    gn = ops.gn_stats(x, gam, bet, 1e-6)
    res_in = torch.randn(B, T, H, W, cout).to(dtype).cuda()
    kw = dict(pad=P2D, pad_mode_t=ZERO, pad_mode_hw=ZERO, prologue=L.PRO_GN_SILU, gn=gn, residual=res_in, gn_out=32)
    one, zero = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    outs = []
    for t_ in (tile, "1x8x32:1x8x1:4"):
        monkeypatch.setenv("CVVAE_CONV_FORCE", t_)
        seen = []
        ops.PROFILE = lambda d, pw_, launch: (seen.append(ops.conv_kernel_name(d)), launch())
        try:
            y, part = ops.conv(x, pw, **kw)
        finally:
            ops.PROFILE = None
        assert seen and ("_t" + t_.split(":")[0] + "_w" + t_.split(":")[1] + "_") in seen[0], seen
        outs.append((y, ops.gn_finalize(part, one, zero, 1e-6)))
    monkeypatch.delenv("CVVAE_CONV_FORCE")
    (ya, ta), (yb, tb) = outs
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    d = (ya.float() - yb.float()).abs()
    assert float((d / yb.float().abs().clamp_min(1.0)).max()) <= 2 * ulp, float(d.max())
    for u, v in zip(ta, tb):
        assert torch.allclose(u, v, rtol=5e-5, atol=5e-6), float((u - v).abs().max())
    a = F.silu(x.float() * gn[0].view(B, 1, 1, 1, cin) + gn[1].view(B, 1, 1, 1, cin)).to(dtype).float().cpu()
    ref = F.conv2d(a.reshape(B * T, H, W, cin).permute(0, 3, 1, 2), w.float()[:, :, 0], b, padding=1)
    ref = ref + res_in.float().cpu().reshape(B * T, H, W, cout).permute(0, 3, 1, 2)
    got = ya.float().cpu().reshape(B * T, H, W, cout).permute(0, 3, 1, 2)
    assert float((got - ref).abs().max()) <= 3 * ulp * float(ref.abs().max()), float((got - ref).abs().max())


def test_starved_grids_get_small_tiles_and_full_grids_keep_theirs(monkeypatch):
    """The instance cost model charges a grid of fewer workgroups than CUs for the CUs it leaves empty (cvvae_api.hip instance_cost):
    a 512-channel per-frame conv at 1x32x32 runs on 64-pixel tiles, the same layer at 17x512^2's 5x64x64 level and the 128-channel
    layers at 256^2 keep the tiles of rounds 1-5; the choice is made per batch item (a batch of clips == single-clip calls, bit for
    bit)."""
    from cvvae_amd import _lib as L
    from cvvae_amd import ops

    def picked(cin, cout, shape, **env):
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        x = torch.randn(*shape, cin).bfloat16().cuda()
        pw = ops.pack_weight((torch.randn(cout, cin, 9) / (cin * 9) ** 0.5).bfloat16().cuda(), torch.zeros(cout).cuda(), (1, 3, 3))
        gn = ops.gn_stats(x, torch.ones(cin).cuda(), torch.zeros(cin).cuda(), 1e-6)
        seen = []
        ops.PROFILE = lambda d, pw_, launch: (seen.append(ops.conv_kernel_name(d)), launch())
        try:
            y = ops.conv(x, pw, pad=P2D, pad_mode_t=ZERO, pad_mode_hw=ZERO, prologue=L.PRO_GN_SILU, gn=gn)
        finally:
            ops.PROFILE = None
            for k_ in env:
                monkeypatch.delenv(k_)
        assert torch.isfinite(y.float()).all()
        return seen[0]

    small = picked(512, 512, (1, 1, 32, 32))
    assert "_t1x2x32_" in small or "_t1x4x32_" in small, small
    batch = picked(512, 512, (4, 1, 32, 32))              # four clips: 4 x 64 workgroups fill the chip -- the per-item choice stays
    assert batch == small, (batch, small)
    big = picked(128, 128, (1, 17, 512, 512))
    assert "_t1x16x32_" in big or "_t1x8x32_w1x4x1_" in big, big    # (the four-wave tile: ops sets cvvae_conv_desc.four_wave)
    forced = picked(512, 512, (1, 1, 32, 32), CVVAE_CONV_FORCE="1x8x32:2x4x1:2")
    assert "_t1x8x32_" in forced, forced


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bfloat16", "float16"])
@pytest.mark.parametrize("tiles", [("1x8x32:1x8x1:2", "1x8x32:1x8x1:1"), ("1x4x32:1x8x1:2", "1x4x32:1x8x1:1")], ids=["256px", "128px"])
def test_32_channel_chunks_of_the_3x3x3_tiles(tiles, dtype, monkeypatch):
    """The BN = 256 tiles of the 3x3x3 conv with 32-channel K chunks (conv_table.h G1, round 6): the same products summed chunk by chunk
    in another order than under 16-channel chunks -- the stored outputs agree to a rounding of the stored type, the fused GroupNorm
    tables to fp32 rounding, causal time folds and tile overhang included; and against fp32 conv3d over the same 16-bit operands.
    (reference op: CausalConv3d, models/vae_blocks3d_sd3.py:16-116)"""
    import torch.nn.functional as F

    from cvvae_amd import _lib as L
    from cvvae_amd import ops
    from tests.test_gpu_grad3d import _pad3
    cin, cout, (B, T, H, W) = 512, 256, (1, 4, 12, 40)
    torch.manual_seed(11)
    x = (torch.randn(B, T, H, W, cin) * 1.5 + 0.3).to(dtype).cuda()
    w = (torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5).to(dtype)
    b = torch.randn(cout) * 0.1
    pw = ops.pack_weight_tfolds(w.cuda(), b.cuda())
    gam, bet = (1.0 + 0.2 * torch.randn(cin)).cuda(), (0.1 * torch.randn(cin)).cuda()
    gn = ops.gn_stats(x, gam, bet, 1e-6)
    kw = dict(pad=PC, pad_mode_t=REP, pad_mode_hw=REP, prologue=L.PRO_GN_SILU, gn=gn, gn_out=32)
    one, zero = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    res = []
    for tile in tiles:
        monkeypatch.setenv("CVVAE_CONV_FORCE", tile)
        seen = []
        ops.PROFILE = lambda d, pw_, launch: (seen.append(ops.conv_kernel_name(d)), launch())
        try:
            y, part = ops.conv(x, pw, **kw)
        finally:
            ops.PROFILE = None
        assert seen and f"_c{16 * int(tile[-1])}_" in seen[0] and ("_t" + tile.split(":")[0] + "_") in seen[0], seen
        res.append((y, ops.gn_finalize(part, one, zero, 1e-6)))
    monkeypatch.delenv("CVVAE_CONV_FORCE")
    (ya, ta), (yb, tb) = res
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    d = (ya.float() - yb.float()).abs()
    assert float((d / yb.float().abs().clamp_min(1.0)).max()) <= 2 * ulp, float(d.max())
    for u, v in zip(ta, tb):
        assert torch.allclose(u, v, rtol=5e-5, atol=5e-6), float((u - v).abs().max())
    a = F.silu(x.float() * gn[0].view(B, 1, 1, 1, cin) + gn[1].view(B, 1, 1, 1, cin)).to(dtype).float().cpu()
    ref = F.conv3d(_pad3(a.permute(0, 4, 1, 2, 3), PC, REP, REP), w.float(), b)
    got = ya.float().cpu().permute(0, 4, 1, 2, 3)
    assert float((got - ref).abs().max()) <= 3 * ulp * float(ref.abs().max()), float((got - ref).abs().max())
