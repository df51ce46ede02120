"""Per-kernel parity on a real MI355X: every C-ABI entry against a plain PyTorch fp32 CPU evaluation of the same
op on the same (16-bit-rounded) inputs.  Tolerances are stated per test."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


# one-rounding error of the storage dtype relative to the data range; fp32: split-precision products + summation order
ULP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 4e-6}
# fast fp32 (one fp16 MFMA + bf8 correction terms): each product is right to ~2^-14; over a dot product the errors add like a
# random walk, so relative to the largest output ~2^-15 is typical -- the band leaves a factor of ~4
FAST_ULP = 1.2e-4


def _ops():
    import cvvae_amd
    from cvvae_amd import ops, _lib
    return ops, _lib


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def to_ndhwc(x):  # [B,C,T,H,W] -> [B,T,H,W,C]
    return x.permute(0, 2, 3, 4, 1).contiguous()


def to_ncdhw(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


def ref_pad(x, pad, mode_t, mode_hw):
    (tf, tb), (hf, hb), (wf, wb) = pad
    if hf or hb or wf or wb:
        x = F.pad(x, (wf, wb, hf, hb, 0, 0), mode="replicate" if mode_hw else "constant")
    if tf or tb:
        x = F.pad(x, (0, 0, 0, 0, tf, tb), mode="replicate" if mode_t else "constant")
    return x


def run_conv_case(dtype, Cin, Cout, k, stride, pad, mode_t, mode_hw, shape, prologue=0, ups=False, out_mode=0,
                  residual=False, seed=0, tol=None, time_folds=False, fast=False, act_bound=None):
    """fast (fp32 only): weights packed for the fast-fp32 kernels (True: CVVAE_F32Q, fp16 MFMA + bf8 correction MFMA; "fp6":
    CVVAE_F32Q6, e3m2 corrections, with act_bound -- default 8 max|gamma| + max|beta|, what engine.WeightCache derives)"""
    ops, L = _ops()
    if dtype == torch.float32 and ups is True:
        pytest.skip("the 27-tap gather form of the upsample conv (CVVAE_FOLD_UPSAMPLE=0 tuning path) has no split-precision instance")
    B, T, H, W = shape
    x = rnd((B, Cin, T, H, W), dtype, seed, 1.0)
    w = rnd((Cout, Cin) + tuple(k), dtype, seed + 1, 1.0 / (Cin * k[0] * k[1] * k[2]) ** 0.5)
    bias = rnd((Cout,), torch.float32, seed + 2, 0.1)
    gamma = 1.0 + rnd((Cin,), torch.float32, seed + 3, 0.1)
    beta = rnd((Cin,), torch.float32, seed + 4, 0.1)
    # ---- fp32 CPU reference on the same rounded inputs
    xr = x.float()
    if prologue:
        xr = F.group_norm(xr, 32, gamma, beta, 1e-6)
        if prologue == 1:
            xr = xr * torch.sigmoid(xr)
    if ups:
        xr = F.interpolate(xr, scale_factor=(1.0, 2.0, 2.0), mode="nearest")
    xr = ref_pad(xr, pad, mode_t, mode_hw)
    ref = F.conv3d(xr, w.float(), bias, stride=stride)
    res = None
    if residual:
        res = rnd(tuple(ref.shape), dtype, seed + 5, 1.0)
        ref = ref + res.float()
    if out_mode == L.OUT_TIME_SHUFFLE:
        b_, nc, t_, h_, w_ = ref.shape
        c = nc // 2
        ref = ref.reshape(b_, 2, c, t_, h_, w_).permute(0, 2, 3, 1, 4, 5).reshape(b_, c, 2 * t_, h_, w_)[:, :, 1:]
    # ---- HIP
    xd = to_ndhwc(x).to(DEV)
    ck = ops.kchunk(k)
    cin_pad = ops.round_up(Cin, ck)
    if cin_pad != Cin:
        xp = torch.zeros(xd.shape[:-1] + (cin_pad,), dtype=dtype, device=DEV)
        xp[..., :Cin] = xd
        xd = xp
    if time_folds:  # packed with the summed time slots for boundary frames (cvvae_pack_weights_tfolds)
        pw = ops.pack_weight_tfolds(w.to(DEV), bias.to(DEV), cin_pad=cin_pad, fast=fast)
    else:
        pw = ops.pack_weight(w.to(DEV), bias.to(DEV), k, cin_pad=cin_pad, fast=fast)
    gn = None
    if prologue:
        g = torch.zeros(cin_pad); g[:Cin] = gamma
        bb = torch.zeros(cin_pad); bb[:Cin] = beta
        assert cin_pad == Cin, "prologue cases use channel counts that need no padding"
        gn = ops.gn_stats(xd, g.to(DEV), bb.to(DEV), 1e-6)
    if ups == 2:
        pw = ops.pack_weight_upfold(w.to(DEV), bias.to(DEV), time_folds=time_folds, fast=fast)
    abd = None
    if fast == "fp6":
        assert pw.dt == L.F32Q6
        if prologue:
            pw.act_bound = float(8.0 * gamma.abs().max() + beta.abs().max()) if act_bound is None else act_bound
        else:  # no GroupNorm in front: the bound is a max-abs reduction on the device (act_bound: a factor on it, for the tests)
            abd = (torch.linalg.vector_norm(xd.reshape(-1), float("inf")) * (1.0 if act_bound is None else act_bound)).reshape(1)
    out = ops.conv(xd, pw, stride=stride, pad=pad, pad_mode_t=mode_t, pad_mode_hw=mode_hw, prologue=prologue, gn=gn,
                   residual=to_ndhwc(res).to(DEV) if residual else None, upsample2x=ups, out_mode=out_mode, act_bound_dev=abd)
    torch.cuda.synchronize()
    got = out.float().cpu()
    if out_mode != L.OUT_NCDHW:
        got = to_ncdhw(got)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    if tol is None:
        # inputs are exact in 16 bit; remaining error = fp32 accumulation order + one output rounding
        # (+ one rounding of the GN/SiLU operand when the prologue is fused)
        base = FAST_ULP if fast else ULP[dtype]
        tol = base * (3.0 if prologue else 1.0)
    assert err <= tol * scale + 1e-6, f"max err {err:.4g} vs scale {scale:.4g} (tol {tol * scale:.4g})"
    return err / scale


# fp32 = split precision (three fp16 MFMAs per product, fp32 tensors): the same cases at ~fp32 accuracy (the torch fp32 CPU
# reference itself carries ~1e-6 of summation-order noise per dot product)
DT = [torch.bfloat16, torch.float16, torch.float32]
REP, ZERO = 1, 0


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Cout", [128, 256, 32, 3])
def test_conv333_sd3_causal(dtype, Cout):
    # sd3 CausalConv3d: replicate W,H (1,1), replicate T front 2 (vae_blocks3d_sd3.py:87-98)
    L = _ops()[1]
    run_conv_case(dtype, 128, Cout, (3, 3, 3), (1, 1, 1), ((2, 0), (1, 1), (1, 1)), REP, REP, (1, 5, 24, 40),
                  out_mode=L.OUT_NCDHW if Cout <= 32 else L.OUT_NDHWC)


@pytest.mark.parametrize("dtype", DT)
def test_conv333_prologue_gn_silu_zero_pad(dtype):
    # vae3d decoder conv: zero pad all faces, fused GroupNorm+SiLU prologue, 256 -> 256
    run_conv_case(dtype, 256, 256, (3, 3, 3), (1, 1, 1), ((1, 1), (1, 1), (1, 1)), ZERO, ZERO, (2, 3, 16, 32), prologue=1)


@pytest.mark.parametrize("dtype", DT)
def test_conv333_prologue_cout128_replicate(dtype):
    run_conv_case(dtype, 128, 128, (3, 3, 3), (1, 1, 1), ((1, 1), (1, 1), (1, 1)), REP, REP, (1, 4, 17, 33), prologue=1)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("T", [17, 3, 16])
def test_conv333_cout128_two_frame_tiles_and_odd_frame_split(dtype, T):
    # 128-channel 3x3x3 layers run two-frame tiles; an odd frame count is split into a two-frame-tile launch over
    # [0, T-1) and a one-frame-tile launch for the last frame (cvvae_api.hip odd_frame_sibling) -- conv numerics of both
    # launches against conv3d, with the fused GroupNorm+SiLU prologue and a residual
    run_conv_case(dtype, 128, 128, (3, 3, 3), (1, 1, 1), ((2, 0), (1, 1), (1, 1)), REP, REP, (1, T, 16, 32), prologue=1,
                  residual=True)


@pytest.mark.parametrize("dtype", DT)
def test_conv333_vae3d_causal_mixed_pad(dtype):
    # vae3d CausalConv3d: zero pad W,H, replicate T front 2 (vae_models.py:301-326)
    run_conv_case(dtype, 128, 256, (3, 3, 3), (1, 1, 1), ((2, 0), (1, 1), (1, 1)), REP, ZERO, (1, 5, 16, 32))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("stride", [(2, 2, 2), (1, 2, 2)])
def test_conv333_downsample_sd3(dtype, stride):
    # sd3 Downsample3D: CausalConv3d stride 2 / (1,2,2), replicate pad (1,1,1,1,2,0)  (vae_blocks3d_sd3.py:203-210)
    run_conv_case(dtype, 128, 128, (3, 3, 3), stride, ((2, 0), (1, 1), (1, 1)), REP, REP, (1, 5, 24, 40))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("stride", [(2, 2, 2), (1, 2, 2)])
def test_conv333_downsample_vae3d(dtype, stride):
    # vae3d Downsample3D: zero pad right/bottom only, replicate T front 2 (vae_models.py:253-260)
    run_conv_case(dtype, 256, 256, (3, 3, 3), stride, ((2, 0), (0, 1), (0, 1)), REP, ZERO, (1, 5, 16, 32))


@pytest.mark.parametrize("dtype", DT)
def test_conv333_upsample_time_shuffle(dtype):
    # sd3 Upsample3D with up_time: nearest x2, replicate conv to 2C, shuffle + drop (vae_blocks3d_sd3.py:342-362)
    L = _ops()[1]
    run_conv_case(dtype, 256, 512, (3, 3, 3), (1, 1, 1), ((1, 1), (1, 1), (1, 1)), REP, REP, (1, 3, 8, 16), ups=True,
                  out_mode=L.OUT_TIME_SHUFFLE)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("mode_hw", [REP, ZERO])
@pytest.mark.parametrize("shuffle", [False, True])
def test_conv333_upsample_folded(dtype, mode_hw, shuffle):
    # Upsample3D as four folded 3x2x2 phase convs (upsample2x=2) against F.interpolate + 27-tap conv3d; odd sizes exercise
    # tile overhang in every phase.  One extra rounding of each folded weight: tolerance 2x the plain conv's.
    L = _ops()[1]
    base = ULP[dtype]
    run_conv_case(dtype, 256, 512, (3, 3, 3), (1, 1, 1), ((1, 1), (1, 1), (1, 1)), REP, mode_hw, (1, 3, 9, 19), ups=2,
                  out_mode=L.OUT_TIME_SHUFFLE if shuffle else L.OUT_NDHWC, tol=2 * base)


TIME_FOLD_CASES = {
    # name: (Cin, Cout, stride, time pad, mode_hw, (B,T,H,W), prologue, ups, shuffle)   -- replicate time padding throughout
    "causal_128_T5": (128, 128, (1, 1, 1), (2, 0), REP, (1, 5, 16, 32), 1, False, False),   # two-frame tiles + odd-frame split
    "causal_128_T2": (128, 128, (1, 1, 1), (2, 0), REP, (2, 2, 9, 33), 1, False, False),    # frames 0 (one slot) and 1 (two)
    "causal_256_T3": (256, 256, (1, 1, 1), (2, 0), REP, (1, 3, 16, 32), 1, False, False),
    "sym_256_T4": (256, 256, (1, 1, 1), (1, 1), REP, (1, 4, 9, 19), 1, False, False),       # first AND last frame fold
    "sym_128_T2": (128, 128, (1, 1, 1), (1, 1), REP, (1, 2, 16, 32), 0, False, False),      # both frames of one two-frame tile
    "sym_512_T1": (512, 512, (1, 1, 1), (1, 1), REP, (1, 1, 8, 32), 0, False, False),       # all three taps on one frame
    "down222_T5": (256, 256, (2, 2, 2), (2, 0), REP, (1, 5, 16, 32), 0, False, False),
    "down122_T4": (128, 128, (1, 2, 2), (2, 0), REP, (1, 4, 16, 32), 0, False, False),
    "out3_T5": (128, 3, (1, 1, 1), (1, 1), REP, (1, 5, 16, 32), 1, False, False),           # conv_out: NCDHW stores
    "upfold_T3": (256, 512, (1, 1, 1), (1, 1), REP, (1, 3, 9, 19), 0, 2, True),
    "upfold_T2_zero_hw": (256, 512, (1, 1, 1), (1, 1), ZERO, (1, 2, 8, 16), 0, 2, False),
}


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", sorted(TIME_FOLD_CASES))
def test_conv333_time_folds(dtype, case):
    # Boundary frames of a clip read one stored frame through two or three time taps (replicate time padding): with the
    # summed weight slots (pack_weight_tfolds) the kernel multiplies that frame once.  Against conv3d on the padded input;
    # one extra rounding of each summed weight: tolerance 2x the plain conv's (x3 with the fused GN+SiLU prologue).
    L = _ops()[1]
    Cin, Cout, stride, tpad, mode_hw, shape, pro, ups, shuffle = TIME_FOLD_CASES[case]
    base = ULP[dtype]
    om = L.OUT_TIME_SHUFFLE if shuffle else (L.OUT_NCDHW if Cout <= 32 else L.OUT_NDHWC)
    run_conv_case(dtype, Cin, Cout, (3, 3, 3), stride, (tpad, (1, 1), (1, 1)), REP, mode_hw, shape, prologue=pro, ups=ups,
                  out_mode=om, tol=2 * base * (3.0 if pro else 1.0), time_folds=True)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("T", [1, 2, 5])
def test_conv333_zero_time_padding_skips_are_exact(dtype, T):
    # zero time padding (vae3d decoder convs): the kernel skips the taps that read a padding frame -- the result must be
    # BIT-identical whether or not the packed weights carry the (unused) time-fold slots, and match conv3d
    ops, L = _ops()
    x = rnd((1, 128, T, 16, 32), dtype, 21, 1.0)
    w = rnd((256, 128, 3, 3, 3), dtype, 22, 1.0 / (128 * 27) ** 0.5)
    bias = rnd((256,), torch.float32, 23, 0.1)
    xd = to_ndhwc(x).to(DEV)
    kw = dict(pad=((1, 1), (1, 1), (1, 1)), pad_mode_t=ZERO, pad_mode_hw=ZERO)
    a = ops.conv(xd, ops.pack_weight(w.to(DEV), bias.to(DEV), (3, 3, 3)), **kw)
    b = ops.conv(xd, ops.pack_weight_tfolds(w.to(DEV), bias.to(DEV)), **kw)
    assert torch.equal(a, b)
    ref = F.conv3d(F.pad(x.float(), (1, 1, 1, 1, 1, 1)), w.float(), bias)
    got = to_ncdhw(a.float().cpu())
    base = ULP[dtype]
    assert (got - ref).abs().max().item() <= base * ref.abs().max().item() + 1e-6


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", ["sum", "center", "sum_down", "sum_up", "center_up_shuffle"])
def test_conv333_single_frame_temporal_fold(dtype, case):
    """Image mode (T = 1): a 3x3x3 conv whose time padding makes its three taps read the one frame (replicate) or zeros
    (zero padding) runs as a 1x3x3 / 1x2x2-phase conv with the coinciding taps summed into the weights
    (cvvae_pack_weights_fold, tfold of cvvae_pack_weights_upfold).  Reference: the unfolded conv3d on the padded frame."""
    ops, L = _ops()
    Cin, Cout = 256, 256
    mode_t = ZERO if case.startswith("center") else REP
    stride = (2, 2, 2) if case == "sum_down" else (1, 1, 1)
    ups = case.endswith("up") or case.endswith("up_shuffle")
    shuffle = case.endswith("shuffle")
    padt = (2, 0) if case == "sum_down" else (1, 1)
    B, H, W = 2, 12, 20
    x = rnd((B, Cin, 1, H, W), dtype, 1, 1.0)
    w = rnd((Cout, Cin, 3, 3, 3), dtype, 2, 1.0 / (Cin * 27) ** 0.5)
    bias = rnd((Cout,), torch.float32, 3, 0.1)
    xr = x.float()
    if ups:
        xr = F.interpolate(xr, scale_factor=(1.0, 2.0, 2.0), mode="nearest")
    xr = ref_pad(xr, (padt, (1, 1), (1, 1)), mode_t, REP)
    ref = F.conv3d(xr, w.float(), bias, stride=stride)
    if shuffle:
        b_, nc, t_, h_, w_ = ref.shape
        ref = ref.reshape(b_, 2, nc // 2, t_, h_, w_).permute(0, 2, 3, 1, 4, 5).reshape(b_, nc // 2, 2 * t_, h_, w_)[:, :, 1:]
    xd = to_ndhwc(x).to(DEV)
    if ups:
        pw = ops.pack_weight_upfold(w.to(DEV), bias.to(DEV), 2 if mode_t == ZERO else 1)
        out = ops.conv(xd, pw, pad=((0, 0), (1, 1), (1, 1)), pad_mode_hw=REP, upsample2x=2,
                       out_mode=L.OUT_TIME_SHUFFLE if shuffle else L.OUT_NDHWC)
    else:
        pw = ops.pack_weight_t1(w.to(DEV), bias.to(DEV), "center" if mode_t == ZERO else "sum")
        out = ops.conv(xd, pw, stride=(1, stride[1], stride[2]), pad=((0, 0), (1, 1), (1, 1)), pad_mode_hw=REP)
    torch.cuda.synchronize()
    got = to_ncdhw(out.float().cpu())
    assert got.shape == ref.shape, (got.shape, ref.shape)
    base = ULP[dtype]
    err = (got - ref).abs().max().item()
    assert err <= 2 * base * ref.abs().max().item() + 1e-6, err


@pytest.mark.parametrize("dtype", DT)
def test_conv333_upsample_vae3d_pad(dtype):
    # vae3d Upsample3D: nearest x2, zero pad W,H, replicate T (1,1), no time upsampling (vae_models.py:218-229)
    run_conv_case(dtype, 256, 256, (3, 3, 3), (1, 1, 1), ((1, 1), (1, 1), (1, 1)), REP, ZERO, (1, 3, 8, 16), ups=True)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Cout", [128, 256])
def test_conv133_prologue_residual(dtype, Cout):
    # ResnetBlock conv2: per-frame 3x3, zero pad 1, GN+SiLU prologue, residual add in the epilogue
    run_conv_case(dtype, Cout, Cout, (1, 3, 3), (1, 1, 1), ((0, 0), (1, 1), (1, 1)), ZERO, ZERO, (1, 3, 24, 40), prologue=1,
                  residual=True)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Cx,C", [(256, 128), (128, 256), (256, 512)])
def test_conv133_fused_shortcut(dtype, Cx, C):
    """ResnetBlock tail in one launch (cvvae_conv_fwd_gn_sc): per-frame 3x3 over GN+SiLU(h) + 1x1 shortcut over x + biases,
    with fused output statistics; reference = conv2d + 1x1 conv + add on the same rounded inputs."""
    ops, L = _ops()
    B, T, H, W = 2, 3, 17, 33
    h = rnd((B, C, T, H, W), dtype, 1, 1.0)
    x = rnd((B, Cx, T, H, W), dtype, 2, 1.0)
    w2 = rnd((C, C, 1, 3, 3), dtype, 3, 1.0 / (C * 9) ** 0.5)
    ws = rnd((C, Cx, 1, 1, 1), dtype, 4, 1.0 / Cx ** 0.5)
    b2, bs = rnd((C,), torch.float32, 5, 0.1), rnd((C,), torch.float32, 6, 0.1)
    gamma, beta = 1.0 + rnd((C,), torch.float32, 7, 0.1), rnd((C,), torch.float32, 8, 0.1)
    ha = F.group_norm(h.float(), 32, gamma, beta, 1e-6)
    ha = ha * torch.sigmoid(ha)
    ref = F.conv3d(F.pad(ha, (1, 1, 1, 1, 0, 0)), w2.float(), b2) + F.conv3d(x.float(), ws.float(), bs)
    hd, xd = to_ndhwc(h).to(DEV), to_ndhwc(x).to(DEV)
    pw2 = ops.pack_weight(w2.reshape(C, C, 9).to(DEV), b2.to(DEV), (1, 3, 3))
    # (fp32 models: both weights accumulate in one register set -> one power-of-two pack scale)
    pws = ops.pack_weight(ws.reshape(C, Cx, 1).to(DEV), bs.to(DEV), (1, 1, 1), wscale=pw2.wscale)
    gn = ops.gn_stats(hd, gamma.to(DEV), beta.to(DEV), 1e-6)
    out, part = ops.conv(hd, pw2, pad=((0, 0), (1, 1), (1, 1)), prologue=L.PRO_GN_SILU, gn=gn, shortcut=(xd, pws),
                         bias=pw2.bias + pws.bias, gn_out=32)
    torch.cuda.synchronize()
    got = to_ncdhw(out.float().cpu())
    base = ULP[dtype]
    err = (got - ref).abs().max().item()
    assert err <= 3 * base * ref.abs().max().item() + 1e-6, err
    one, zero = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    sc_f, sh_f = ops.gn_finalize(part, one, zero, 1e-6)
    sc_s, sh_s = ops.gn_stats(out, one, zero, 1e-6)
    assert (sc_f - sc_s).abs().max().item() <= 2e-5 * sc_s.abs().max().item()
    assert (sh_f - sh_s).abs().max().item() <= 2e-5 * max(sh_s.abs().max().item(), 1.0)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Cout", [256, 128])
def test_conv111_shortcut(dtype, Cout):
    run_conv_case(dtype, 128, Cout, (1, 1, 1), (1, 1, 1), ((0, 0), (0, 0), (0, 0)), ZERO, ZERO, (1, 3, 10, 52))


@pytest.mark.parametrize("dtype", DT)
def test_conv111_gn_only_prologue(dtype):
    run_conv_case(dtype, 512, 512, (1, 1, 1), (1, 1, 1), ((0, 0), (0, 0), (0, 0)), ZERO, ZERO, (1, 1, 1, 300), prologue=2)


@pytest.mark.parametrize("dtype", DT)
def test_conv_small_cin_padded(dtype):
    # conv_in: 3 input channels padded to the 16-channel chunk with zero weights
    run_conv_case(dtype, 3, 128, (3, 3, 3), (1, 1, 1), ((2, 0), (1, 1), (1, 1)), REP, REP, (1, 5, 16, 40))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", ["c333_128", "c133_256_res", "ups_shuffle_512", "upfold_shuffle_512", "upfold_256", "c111_512_b2", "down_222",
                                  "c333_128_t2"])
def test_conv_fused_gn_stats(dtype, case):
    """cvvae_conv_fwd_gn + cvvae_gn_finalize (statistics from the conv epilogue) must equal a statistics pass over the
    stored output (cvvae_gn_stats) and torch's group_norm moments of it: every record written once, all layouts."""
    ops, L = _ops()
    cfg = {
        # Cin, Cout, k, stride, pad, shape(B,T,H,W), ups, out_mode, residual
        "c333_128": (128, 128, (3, 3, 3), (1, 1, 1), ((2, 0), (1, 1), (1, 1)), (1, 5, 24, 40), False, L.OUT_NDHWC, False),
        "c333_128_t2": (128, 128, (3, 3, 3), (1, 1, 1), ((1, 1), (1, 1), (1, 1)), (2, 17, 16, 32), False, L.OUT_NDHWC, False),
        "c133_256_res": (256, 256, (1, 3, 3), (1, 1, 1), ((0, 0), (1, 1), (1, 1)), (1, 3, 17, 33), False, L.OUT_NDHWC, True),
        "ups_shuffle_512": (256, 512, (3, 3, 3), (1, 1, 1), ((1, 1), (1, 1), (1, 1)), (1, 3, 8, 16), True, L.OUT_TIME_SHUFFLE, False),
        "upfold_shuffle_512": (256, 512, (3, 3, 3), (1, 1, 1), ((1, 1), (1, 1), (1, 1)), (1, 3, 9, 19), 2, L.OUT_TIME_SHUFFLE, False),
        "upfold_256": (256, 256, (3, 3, 3), (1, 1, 1), ((1, 1), (1, 1), (1, 1)), (2, 2, 8, 16), 2, L.OUT_NDHWC, False),
        "c111_512_b2": (256, 512, (1, 1, 1), (1, 1, 1), ((0, 0), (0, 0), (0, 0)), (2, 1, 1, 300), False, L.OUT_NDHWC, True),
        "down_222": (128, 128, (3, 3, 3), (2, 2, 2), ((2, 0), (1, 1), (1, 1)), (1, 5, 24, 40), False, L.OUT_NDHWC, False),
    }[case]
    Cin, Cout, k, stride, pad, (B, T, H, W), ups, out_mode, residual = cfg
    if dtype == torch.float32 and ups is True:
        pytest.skip("no split-precision instance of the 27-tap gather form of the upsample conv")
    x = to_ndhwc(rnd((B, Cin, T, H, W), dtype, 1, 1.0)).to(DEV)
    w = rnd((Cout, Cin) + k, dtype, 2, 1.0 / (Cin * k[0] * k[1] * k[2]) ** 0.5)
    bias = rnd((Cout,), torch.float32, 3, 0.5)  # a non-zero mean makes the (n, mean, M2) merge matter
    pw = ops.pack_weight_upfold(w.to(DEV), bias.to(DEV)) if ups == 2 else ops.pack_weight(w.to(DEV), bias.to(DEV), k)
    y0 = ops.conv(x, pw, stride=stride, pad=pad, pad_mode_t=REP, pad_mode_hw=REP, upsample2x=ups, out_mode=out_mode)
    res = rnd(tuple(y0.shape), dtype, 4, 1.0).to(DEV) if residual else None
    y, part = ops.conv(x, pw, stride=stride, pad=pad, pad_mode_t=REP, pad_mode_hw=REP, upsample2x=ups, out_mode=out_mode,
                       residual=res, gn_out=32)
    C = y.shape[-1]
    gamma = (1.0 + rnd((C,), torch.float32, 5, 0.1)).to(DEV)
    beta = rnd((C,), torch.float32, 6, 0.1).to(DEV)
    sc_f, sh_f = ops.gn_finalize(part, gamma, beta, 1e-6)
    sc_s, sh_s = ops.gn_stats(y, gamma, beta, 1e-6)
    torch.cuda.synchronize()
    yr = to_ncdhw(y.float().cpu())
    mean = yr.reshape(B, 32, -1).mean(-1)
    var = yr.reshape(B, 32, -1).var(-1, unbiased=False)
    rstd = (var + 1e-6).rsqrt()
    cpg = C // 32
    sc_ref = gamma.cpu().reshape(1, 32, cpg) * rstd.reshape(B, 32, 1)
    sh_ref = beta.cpu().reshape(1, 32, cpg) - mean.reshape(B, 32, 1) * sc_ref
    for got_sc, got_sh, what in ((sc_f, sh_f, "fused"), (sc_s, sh_s, "pass")):
        e1 = (got_sc.cpu().reshape(B, 32, cpg) - sc_ref).abs().max().item() / sc_ref.abs().max().item()
        e2 = (got_sh.cpu().reshape(B, 32, cpg) - sh_ref).abs().max().item() / max(sh_ref.abs().max().item(), 1.0)
        assert e1 < 2e-5 and e2 < 2e-5, (what, e1, e2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("per_frame", [False, True])
def test_gn_stats(dtype, per_frame):
    ops, L = _ops()
    B, C, T, H, W = 2, 256, 3, 12, 20
    x = rnd((B, C, T, H, W), dtype, 0, 2.0) + 3.0  # offset mean stresses the variance algorithm
    x = x.to(dtype)
    gamma = 1.0 + rnd((C,), torch.float32, 1, 0.1)
    beta = rnd((C,), torch.float32, 2, 0.1)
    sc, sh = ops.gn_stats(to_ndhwc(x).to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6, per_frame=per_frame)
    xf = x.float()
    if per_frame:
        xf = xf.permute(0, 2, 1, 3, 4).reshape(B * T, C, H * W)
    else:
        xf = xf.reshape(B, C, T * H * W)
    rows = xf.shape[0]
    g = xf.reshape(rows, 32, -1).double()
    mean = g.mean(-1)
    var = g.var(-1, unbiased=False)
    rstd = (var + 1e-6).rsqrt()
    sc_ref = (gamma.double().view(1, 32, -1) * rstd.unsqueeze(-1)).reshape(rows, C)
    sh_ref = beta.double().view(1, C) - (mean.unsqueeze(-1).expand(rows, 32, C // 32).reshape(rows, C)) * sc_ref
    assert (sc.cpu().double() - sc_ref).abs().max() <= 1e-5 * sc_ref.abs().max()
    assert (sh.cpu().double() - sh_ref).abs().max() <= 2e-5 * max(1.0, sh_ref.abs().max().item())


@pytest.mark.parametrize("dtype", DT)
def test_softmax_transpose_layernorm_tattn(dtype):
    ops, L = _ops()
    s = rnd((37, 200), torch.float32, 0, 3.0)
    sp = torch.zeros(37, 256); sp[:, :200] = s; sp[:, 200:] = 99.0  # garbage in the padding must be ignored
    p = ops.softmax_rows(sp.to(DEV), 200, dtype)
    ref = torch.softmax(s, -1)
    tol = ULP[dtype]
    assert (p[:, :200].float().cpu() - ref).abs().max() <= tol * ref.max() + 1e-6
    assert p[:, 200:].float().abs().max().item() == 0.0
    x = rnd((3, 70, 45), dtype, 1)
    assert torch.equal(ops.transpose(x.to(DEV)).cpu(), x.transpose(1, 2).contiguous())
    y = rnd((50, 512), dtype, 2, 2.0)
    gam = 1.0 + rnd((512,), torch.float32, 3, 0.1)
    bet = rnd((512,), torch.float32, 4, 0.1)
    ln = ops.layernorm(y.to(DEV), gam.to(DEV), bet.to(DEV), 1e-5).float().cpu()
    lref = F.layer_norm(y.float(), (512,), gam, bet, 1e-5)
    assert (ln - lref).abs().max() <= tol * lref.abs().max() * 1.5
    # temporal attention straight on NDHWC [B,T,H,W,C]; reference = '(b h w) t c' attention (vae_models.py:626-628)
    q, k, v = rnd((2, 5, 3, 11, 512), dtype, 5), rnd((2, 5, 3, 11, 512), dtype, 6), rnd((2, 5, 3, 11, 512), dtype, 7)
    o = ops.temporal_attention(q.to(DEV), k.to(DEV), v.to(DEV)).float().cpu()
    tok = lambda t: t.float().permute(0, 2, 3, 1, 4).reshape(-1, 5, 512)  # noqa: E731
    sc = torch.softmax(tok(q) @ tok(k).transpose(1, 2) * 512 ** -0.5, -1)
    oref = (sc @ tok(v)).reshape(2, 3, 11, 5, 512).permute(0, 3, 1, 2, 4)
    assert (o - oref).abs().max() <= tol * oref.abs().max() * 1.5


@pytest.mark.parametrize("dtype", DT)
def test_layout_and_blend(dtype):
    ops, L = _ops()
    x = rnd((2, 3, 4, 9, 11), torch.float32, 0)
    nd = ops.ncdhw_to_ndhwc(x.to(DEV), 16, dtype)
    assert nd.shape == (2, 4, 9, 11, 16)
    assert torch.equal(nd[..., :3].cpu(), to_ndhwc(x.to(dtype)))
    assert nd[..., 3:].float().abs().max().item() == 0.0
    back = ops.ndhwc_to_ncdhw(nd, 3)
    assert torch.equal(back.cpu(), x.to(dtype))
    for axis in (0, 1):
        a = rnd((1, 4, 3, 20, 24), dtype, 1)
        b = rnd((1, 4, 3, 20, 24), dtype, 2)
        o = 6
        wgt = torch.arange(o) / o
        ref = b.clone()
        if axis == 0:
            w5 = wgt.view(1, 1, 1, -1, 1)
            ref[:, :, :, :o, :] = ((1 - w5) * a[:, :, :, -o:, :] + w5 * b[:, :, :, :o, :]).to(dtype)
        else:
            w5 = wgt.view(1, 1, 1, 1, -1)
            ref[:, :, :, :, :o] = ((1 - w5) * a[:, :, :, :, -o:] + w5 * b[:, :, :, :, :o]).to(dtype)
        got = ops.blend_(a.to(DEV), b.to(DEV).clone(), o, axis).cpu()
        # same fp32 formula; an FMA contraction can move one value by one 16-bit ulp
        assert (got.float() - ref.float()).abs().max() <= ULP[dtype] * 8


def test_full_size_conv_properties(monkeypatch):
    """BASELINE config 3's largest layer shape (128 -> 128, 3x3x3 causal, [1,17,512,512]): size-independent properties instead
    of an oracle.  (a) two different tilings of the same launch (the library's choice: two-frame tiles, odd-frame split, short
    tiles last -- and a forced one-frame tile) accumulate every output in the same order, so they must agree BIT FOR BIT: any
    error in the block -> tile map, the halo addressing or the time-fold plan of either shows up; (b) linearity: conv(2x) ==
    2 conv(x) exactly (scaling by 2 commutes with every rounding); (c) the fused GroupNorm statistics of the stored tensor
    agree with a statistics pass over it."""
    ops, L = _ops()
    dtype = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(5)
    x = (torch.rand((1, 17, 512, 512, 128), generator=g) * 2 - 1).to(dtype).to(DEV)
    w = ((torch.rand((128, 128, 3, 3, 3), generator=g) * 2 - 1) / (128 * 27) ** 0.5).to(dtype).to(DEV)
    pw = ops.pack_weight_tfolds(w, None)
    kw = dict(pad=((2, 0), (1, 1), (1, 1)), pad_mode_t=REP, pad_mode_hw=REP)
    y, part = ops.conv(x, pw, gn_out=32, **kw)
    name = {}
    ops.PROFILE = lambda d, pw_, launch: (name.setdefault("k", ops.conv_kernel_name(d)), launch())
    monkeypatch.setenv("CVVAE_CONV_FORCE", "1x8x32:2x4x1:1")
    y1 = ops.conv(x, pw, **kw)
    ops.PROFILE = None
    monkeypatch.delenv("CVVAE_CONV_FORCE")
    assert "t1x8x32" in name["k"], name
    assert torch.equal(y, y1), "two tilings of the same conv differ"
    y2 = ops.conv(x * 2, pw, **kw)
    assert torch.equal(y2, y * 2), "conv(2x) != 2 conv(x)"
    ones, zeros = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    sc_f, sh_f = ops.gn_finalize(part, ones, zeros, 1e-6)
    sc_s, sh_s = ops.gn_stats(y, ones, zeros, 1e-6)
    assert torch.allclose(sc_f, sc_s, rtol=2e-4, atol=1e-6) and torch.allclose(sh_f, sh_s, rtol=2e-4, atol=2e-5)


def test_full_size_upsample_conv_properties():
    """The 16.7-TFLOP Upsample3D conv of BASELINE config 3 (256 -> 512 at [1,9,256,256], nearest-2x + 3x3x3 + channel->time
    shuffle, as four folded phase convs with time folds): (a) linearity, bit-exact; (b) translation: the same conv over a crop
    of the input rows reproduces the corresponding output rows bit for bit away from the crop's padding -- every output is
    accumulated in the same order wherever its tile lies, so a wrong tile / phase / halo address shows up."""
    ops, L = _ops()
    dtype = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(6)
    x = (torch.rand((1, 9, 256, 256, 256), generator=g) * 2 - 1).to(dtype).to(DEV)
    w = ((torch.rand((512, 256, 3, 3, 3), generator=g) * 2 - 1) / (256 * 27) ** 0.5).to(dtype).to(DEV)
    pw = ops.pack_weight_upfold(w, None, time_folds=True)
    kw = dict(pad=((1, 1), (1, 1), (1, 1)), pad_mode_t=REP, pad_mode_hw=REP, upsample2x=2, out_mode=L.OUT_TIME_SHUFFLE)
    y = ops.conv(x, pw, **kw)
    assert tuple(y.shape) == (1, 17, 512, 512, 256)
    assert torch.equal(ops.conv(x * 2, pw, **kw), y * 2), "conv(2x) != 2 conv(x)"
    yc = ops.conv(x[:, :, 64:192].contiguous(), pw, **kw)  # stored rows [64, 192) -> output rows [128, 384)
    assert torch.equal(yc[:, :, 2:-2], y[:, :, 130:382]), "a crop of the input does not reproduce the rows of the full output"
