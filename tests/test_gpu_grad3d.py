"""Training the 3-D encoder on the kernels (cvvae_amd/grad3d.py; /root/reference/lvdm/models/autoencoder.py:1057-1090 runs
`z, xrec = self(x)` under autograd): the weight-gradient kernel, the bias / GroupNorm-affine reductions and the padding adjoint
against torch autograd of the same op, then a whole (small) sd3 encoder -- conv_in, a down block of every kind (2x2x2 and 1x2x2
Downsample3D), the mid block with its attention, norm_out + conv_out -- against autograd over the ORACLE's ops (plain PyTorch fp32 on
the CPU, same dtype-rounded weights and inputs), and the training step's chain loss(constraint_decoder(encoder(x))).backward().
Errors are relative L2 norms; every printed value is appended to gpurun_out/grad3d_parity.txt (kept as profiles/r4_grad3d_parity.log)."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import cvvae_oracle as O
from oracle.seeded import seeded_input, seeded_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = [torch.float32, torch.float16, torch.bfloat16]
# one op: the operands are exact in the storage dtype, the products accumulate in fp32: only the summation order differs
# (fp32 models: three bf16 MFMAs per product, ~2^-16 relative)
WGRAD_TOL = {torch.float32: 5e-5, torch.float16: 2e-5, torch.bfloat16: 2e-5}
NET_IN_TOL = {torch.float32: 2e-3, torch.float16: 2e-2, torch.bfloat16: 1e-1}     # input gradient through 11 blocks
NET_W_TOL = {torch.float32: 3e-3, torch.float16: 3e-2, torch.bfloat16: 1.5e-1}    # worst parameter tensor
REP, ZERO = 1, 0


def rel(a, b, floor=0.0):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(floor if floor else 1e-30))


def _log(line):
    print("\n" + line)
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "grad3d_parity.txt"), "a") as f:
            f.write(line + "\n")


def _pad3(f, pad, mode_t, mode_hw):
    (tf, tb), (hf, hb), (wf, wb) = pad
    if hf or hb or wf or wb:
        f = F.pad(f, (wf, wb, hf, hb, 0, 0), mode="replicate") if mode_hw == REP else F.pad(f, (wf, wb, hf, hb))
    if tf or tb:
        f = F.pad(f, (0, 0, 0, 0, tf, tb), mode="replicate") if mode_t == REP else F.pad(f, (0, 0, 0, 0, tf, tb))
    return f


CASES = [
    # name, Cin (real), Cs (stored), Cout (real), Cg (stored), k, stride, pad, mode_t, mode_hw, (B, T, H, W)
    ("causal333_128", 128, 128, 128, 128, (3, 3, 3), (1, 1, 1), ((2, 0), (1, 1), (1, 1)), REP, REP, (1, 5, 12, 70)),
    ("sym333_256to128", 256, 256, 128, 128, (3, 3, 3), (1, 1, 1), ((1, 1), (1, 1), (1, 1)), REP, REP, (2, 3, 9, 33)),
    ("zero333_128to256", 128, 128, 256, 256, (3, 3, 3), (1, 1, 1), ((1, 1), (1, 1), (1, 1)), ZERO, ZERO, (1, 2, 8, 64)),
    ("down222", 128, 128, 128, 128, (3, 3, 3), (2, 2, 2), ((2, 0), (1, 1), (1, 1)), REP, REP, (1, 5, 16, 40)),
    ("down122", 256, 256, 256, 256, (3, 3, 3), (1, 2, 2), ((2, 0), (1, 1), (1, 1)), REP, REP, (1, 3, 10, 24)),
    ("frame133_zero", 128, 128, 256, 256, (1, 3, 3), (1, 1, 1), ((0, 0), (1, 1), (1, 1)), ZERO, ZERO, (2, 3, 20, 36)),
    ("conv_in_3of16", 3, 16, 128, 128, (3, 3, 3), (1, 1, 1), ((2, 0), (1, 1), (1, 1)), REP, REP, (1, 5, 24, 40)),
    ("conv_out_32", 512, 512, 32, 32, (3, 3, 3), (1, 1, 1), ((2, 0), (1, 1), (1, 1)), REP, REP, (1, 3, 8, 12)),
    ("linear_1x1", 512, 512, 512, 512, (1, 1, 1), (1, 1, 1), ((0, 0), (0, 0), (0, 0)), ZERO, ZERO, (3, 1, 1, 150)),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_wgrad(case, dtype):
    """cvvae_conv_wgrad vs autograd's weight gradient of F.conv3d over the padded operand (every padding flavour, both strides,
    channel padding on either side, row lengths that are not a multiple of the 64-pixel K panel); bit-reproducible."""
    from cvvae_amd import ops
    name, cin, cs, cout, cg, k, stride, pad, mt, mhw, (B, T, H, W) = case
    torch.manual_seed(len(name))
    a = torch.zeros(B, T, H, W, cs)
    a[..., :cin] = torch.randn(B, T, H, W, cin)
    a = a.to(dtype)
    f = _pad3(a.float()[..., :cin].permute(0, 4, 1, 2, 3), pad, mt, mhw)
    w = torch.zeros(cout, cin, *k, requires_grad=True)
    y = F.conv3d(f, w, None, stride=stride)
    gy = torch.zeros(B, *y.shape[2:], cg)
    gy[..., :cout] = torch.randn(B, *y.shape[2:], cout)
    gy = gy.to(dtype)
    (y * gy.float()[..., :cout].permute(0, 4, 1, 2, 3)).sum().backward()
    got = ops.conv_wgrad(a.cuda(), gy.cuda(), k, stride=stride, pad=pad, pad_mode_t=mt, pad_mode_hw=mhw, cin=cin, cout=cout)
    e = rel(got, w.grad)
    _log(f"[wgrad {name} {str(dtype)[6:]}] dW {tuple(got.shape)} rel {e:.2e}")
    assert got.shape == w.shape and got.dtype == torch.float32
    assert e <= WGRAD_TOL[dtype]
    again = ops.conv_wgrad(a.cuda(), gy.cuda(), k, stride=stride, pad=pad, pad_mode_t=mt, pad_mode_hw=mhw, cin=cin, cout=cout)
    assert torch.equal(got, again), "not deterministic"
    # the bias gradient out of the same launch (ABI 12: fused by the 16-bit 3x3 kernel, cvvae_channel_sums elsewhere)
    dw2, db = ops.conv_wgrad(a.cuda(), gy.cuda(), k, stride=stride, pad=pad, pad_mode_t=mt, pad_mode_hw=mhw, cin=cin, cout=cout, bias=True)
    want = gy.float()[..., :cout].reshape(-1, cout).sum(0)
    eb = rel(db, want)
    _log(f"[wgrad {name} {str(dtype)[6:]}] db rel {eb:.2e}")
    assert torch.equal(dw2, got) and db.shape == (cout,) and db.dtype == torch.float32 and eb <= 5e-5


@pytest.mark.parametrize("dtype", DT)
def test_bias_and_groupnorm_affine_gradients(dtype):
    from cvvae_amd import ops
    torch.manual_seed(3)
    for (B, T, H, W, C, per_frame, silu) in [(2, 3, 40, 52, 128, False, True), (1, 5, 9, 7, 512, True, False), (2, 1, 33, 17, 256, False, True)]:
        x = (torch.randn(B, T, H, W, C) * 1.5 + 1.0).to(dtype)
        gy = torch.randn(B, T, H, W, C).to(dtype)
        gamma = (torch.randn(C) * 0.5 + 1.0).requires_grad_(True)
        beta = (torch.randn(C) * 0.3).requires_grad_(True)
        xf = x.float()
        f = xf.permute(0, 1, 4, 2, 3).reshape(B * T, C, H * W) if per_frame else xf.permute(0, 4, 1, 2, 3).reshape(B, C, -1)
        y = F.group_norm(f, 32, gamma, beta, 1e-6)
        y = F.silu(y) if silu else y
        gyf = gy.float().permute(0, 1, 4, 2, 3).reshape(B * T, C, H * W) if per_frame else gy.float().permute(0, 4, 1, 2, 3).reshape(B, C, -1)
        y.backward(gyf)
        xd = x.cuda()
        one, zero = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        tabs = ops.gn_stats(xd, one, zero, 1e-6, per_frame=per_frame)
        dg, db = ops.gn_bwd_params(xd, gy.cuda(), tabs, gamma.detach().cuda(), beta.detach().cuda(), silu, per_frame=per_frame)
        bg = ops.bias_grad(gy.cuda())
        # the fused form (affine sums on the input gradient's reduction pass) against the separate kernels
        gx_ref = ops.gn_bwd_input(xd, gy.cuda(), tabs, gamma.detach().cuda(), beta.detach().cuda(), silu, per_frame=per_frame)
        gx2, dg2, db2 = ops.gn_bwd_input_params(xd, gy.cuda(), tabs, gamma.detach().cuda(), beta.detach().cuda(), silu, per_frame=per_frame)
        assert torch.equal(gx2, gx_ref) and rel(dg2, dg) <= 1e-5 and rel(db2, db) <= 1e-5, (rel(dg2, dg), rel(db2, db))
        assert rel(dg2, gamma.grad) <= 2e-4 and rel(db2, beta.grad) <= 2e-4
        e = (rel(dg, gamma.grad), rel(db, beta.grad), rel(bg, gy.float().reshape(-1, C).sum(0)))
        _log(f"[affine grads {str(dtype)[6:]} C{C} {T}x{H}x{W} per_frame={per_frame}] d gamma {e[0]:.2e} d beta {e[1]:.2e} bias {e[2]:.2e}")
        assert max(e) <= 2e-4, e  # (fp32 sums of 16-bit-exact products; GroupNorm statistics from the kernel's own pass)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("pad_t,mode", [((2, 0), REP), ((1, 1), REP), ((1, 1), ZERO)])
def test_pad_fold_is_the_adjoint_of_the_padding(dtype, pad_t, mode):
    from cvvae_amd import ops
    torch.manual_seed(5)
    B, T, H, W, C = 2, 3, 6, 7, 16
    gp = torch.randn(B, T + pad_t[0] + pad_t[1], H + 2, W + 2, C).to(dtype)
    add = torch.randn(B, T, H, W, C).to(dtype)
    x = torch.zeros(B, C, T, H, W, requires_grad=True)
    y = _pad3(x, (pad_t, (1, 1), (1, 1)), mode, mode)
    (y * gp.float().permute(0, 4, 1, 2, 3)).sum().backward()
    ref = x.grad.permute(0, 2, 3, 4, 1) + add.float()
    got = ops.pad_fold(gp.cuda(), pad_t, 1, mode, mode, add=add.cuda())
    assert rel(got, ref) <= {torch.float32: 1e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_temporal_attention_and_layernorm_backward(dtype):
    """cvvae_temporal_attention_bwd (per pixel over T <= 8 frames) and the LayerNorm backward on the GroupNorm kernels (one group,
    rows = tokens) against autograd"""
    from cvvae_amd import ops
    torch.manual_seed(9)
    tol = {torch.float32: 2e-5, torch.float16: 3e-3, torch.bfloat16: 2e-2}[dtype]
    for (B, T, H, W, C) in [(1, 5, 6, 7, 512), (2, 8, 3, 4, 128), (1, 1, 4, 4, 256)]:
        q, k, v, go = [torch.randn(B, T, H, W, C).to(dtype) for _ in range(4)]
        t = [a.float().permute(0, 2, 3, 1, 4).reshape(-1, T, C).clone().requires_grad_(True) for a in (q, k, v)]
        p = torch.softmax(t[0] @ t[1].transpose(1, 2) * (C ** -0.5), -1)
        (p @ t[2] * go.float().permute(0, 2, 3, 1, 4).reshape(-1, T, C)).sum().backward()
        ref = [a.grad.reshape(B, H, W, T, C).permute(0, 3, 1, 2, 4) for a in t]
        got = ops.temporal_attention_bwd(q.cuda(), k.cuda(), v.cuda(), go.cuda())
        e = [rel(a, b) for a, b in zip(got, ref)]
        _log(f"[temporal attention bwd {str(dtype)[6:]} T{T} {H}x{W} C{C}] gq {e[0]:.2e} gk {e[1]:.2e} gv {e[2]:.2e}")
        assert max(e) <= tol, e
        x = (torch.randn(B, T, H, W, C) * 1.3 + 0.5).to(dtype)
        gamma, beta = (torch.randn(C) * 0.4 + 1.0).requires_grad_(True), (torch.randn(C) * 0.2).requires_grad_(True)
        xr = x.float().clone().requires_grad_(True)
        (F.layer_norm(xr, (C,), gamma, beta, 1e-5) * go.float()).sum().backward()
        gx, dg, db = ops.layernorm_bwd(x.cuda(), go.cuda(), gamma.detach().cuda(), beta.detach().cuda(), 1e-5)
        e = (rel(gx, xr.grad), rel(dg, gamma.grad), rel(db, beta.grad))
        _log(f"[layernorm bwd {str(dtype)[6:]} tokens {B * T * H * W} C{C}] gx {e[0]:.2e} d gamma {e[1]:.2e} d beta {e[2]:.2e}")
        assert e[0] <= tol and max(e[1:]) <= 2e-4, e


SMALL = dict(block_out_channels=[128, 256, 512], layers_per_block=1)


def _encoder(dtype):
    import cvvae_amd
    m = cvvae_amd.CVVAESD3Model(**SMALL)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 7)
    m.load_state_dict(sd, strict=True)
    m = m.to(dtype).cuda()
    ref_sd = {k: v.to(dtype).float().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("encoder.")}
    return m, ref_sd


@pytest.mark.parametrize("dtype", DT)
def test_sd3_encoder_backward_vs_autograd_of_the_oracle(dtype):
    """conv_in -> down block (ResnetBlock3D + 2x2x2 Downsample3D) -> down block (channel change: 1x1 shortcut; 1x2x2 Downsample3D)
    -> block without downsampler -> mid block (ResnetBlock3D, spatial attention at 512 channels, ResnetBlock3D) -> norm_out ->
    conv_out: input gradient and EVERY parameter gradient against autograd over the oracle's ops; and the result is reproducible."""
    m, ref_sd = _encoder(dtype)
    enc = m.encoder.train()
    x = seeded_input((1, 3, 5, 32, 32), 11).to(dtype)
    xr = x.float().clone().requires_grad_(True)
    yr = O.sd3_encoder(xr, ref_sd, dict(SMALL))
    cot = seeded_input(tuple(yr.shape), 4).to(dtype)
    (yr * cot.float()).sum().backward()
    xa = x.cuda().requires_grad_(True)
    ya = enc(xa)
    assert ya.requires_grad
    (ya.float() * cot.cuda().float()).sum().backward()
    e_y, e_x = rel(ya, yr), rel(xa.grad, xr.grad)
    names = [n for n, _ in enc.named_parameters()]
    scale = max(float(ref_sd["encoder." + n].grad.norm()) for n in names)
    errs = sorted(((rel(p.grad, ref_sd["encoder." + n].grad, 1e-3 * scale), n) for n, p in enc.named_parameters()), reverse=True)
    conv_w = [e for e, n in errs if n.endswith("weight") and ("conv" in n or "to_" in n)]
    _log(f"[sd3 encoder backward {str(dtype)[6:]}] forward rel {e_y:.2e}; dL/dx rel {e_x:.2e}; parameters: worst {errs[0][0]:.2e} "
         f"({errs[0][1]}), median {errs[len(errs) // 2][0]:.2e}, conv/linear weights worst {max(conv_w):.2e} ({len(names)} tensors)")
    assert e_x <= NET_IN_TOL[dtype], e_x
    assert errs[0][0] <= NET_W_TOL[dtype], errs[:5]
    assert all(p.grad is not None and p.grad.dtype == p.dtype for p in enc.parameters())
    # reproducible: the same step again gives the same bits
    g1 = {n: p.grad.clone() for n, p in enc.named_parameters()}
    enc.zero_grad()
    xb = x.cuda().requires_grad_(True)
    (enc(xb).float() * cot.cuda().float()).sum().backward()
    assert torch.equal(xb.grad, xa.grad) and all(torch.equal(p.grad, g1[n]) for n, p in enc.named_parameters())


@pytest.mark.parametrize("dtype", DT)
def test_sd3_decoder_backward_vs_autograd_of_the_oracle(dtype):
    """Decoder3D: conv_in over the latent, mid block (attention at 512 channels), up blocks with both Upsample3D kinds (time
    shuffle + frame drop; spatial only), norm_out + conv_out: dL/dz and every parameter gradient against autograd over the oracle"""
    import cvvae_amd
    m = cvvae_amd.CVVAESD3Model(**SMALL)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 8)
    m.load_state_dict(sd, strict=True)
    m = m.to(dtype).cuda()
    dec = m.decoder.train()
    ref_sd = {k: v.to(dtype).float().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("decoder.")}
    z = seeded_input((1, 16, 3, 8, 8), 13).to(dtype)
    zr = z.float().clone().requires_grad_(True)
    yr = O.sd3_decoder(zr, ref_sd, dict(SMALL))
    cot = seeded_input(tuple(yr.shape), 5).to(dtype)
    (yr * cot.float()).sum().backward()
    za = z.cuda().requires_grad_(True)
    ya = dec(za)
    (ya.float() * cot.cuda().float()).sum().backward()
    e_y, e_z = rel(ya, yr), rel(za.grad, zr.grad)
    names = [n for n, _ in dec.named_parameters()]
    scale = max(float(ref_sd["decoder." + n].grad.norm()) for n in names)
    errs = sorted(((rel(p.grad, ref_sd["decoder." + n].grad, 1e-3 * scale), n) for n, p in dec.named_parameters()), reverse=True)
    _log(f"[sd3 decoder backward {str(dtype)[6:]}] forward rel {e_y:.2e}; dL/dz rel {e_z:.2e}; parameters: worst {errs[0][0]:.2e} "
         f"({errs[0][1]}), median {errs[len(errs) // 2][0]:.2e} ({len(names)} tensors)")
    assert e_z <= NET_IN_TOL[dtype], e_z
    assert errs[0][0] <= NET_W_TOL[dtype], errs[:5]


@pytest.mark.parametrize("dtype", DT)
def test_vae3d_encoder_backward_vs_autograd_of_the_oracle(dtype):
    """the SD2.1-compatible family's Encoder (vae_models.py:790-823) through the same tape walker: causal convs with zero H / W
    padding, Downsample3D's (0, 1) pads in both stride kinds, GroupNorm eps 1e-5, nin_shortcut, the mid block's attention"""
    import cvvae_amd
    over = dict(ch=128, ch_mult=(1, 2, 4), num_res_blocks=1)
    m = cvvae_amd.CVVAEModel(**over)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 9)
    m.load_state_dict(sd, strict=True)
    m = m.to(dtype).cuda()
    enc = m.encoder.train()
    ref_sd = {k: v.to(dtype).float().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("encoder.")}
    x = seeded_input((1, 3, 5, 32, 32), 14).to(dtype)
    xr = x.float().clone().requires_grad_(True)
    yr = O.v3_encoder(xr, ref_sd, dict(over))
    cot = seeded_input(tuple(yr.shape), 6).to(dtype)
    (yr * cot.float()).sum().backward()
    xa = x.cuda().requires_grad_(True)
    ya = enc(xa)
    (ya.float() * cot.cuda().float()).sum().backward()
    e_x = rel(xa.grad, xr.grad)
    names = [n for n, _ in enc.named_parameters()]
    scale = max(float(ref_sd["encoder." + n].grad.norm()) for n in names)
    errs = sorted(((rel(p.grad, ref_sd["encoder." + n].grad, 1e-3 * scale), n) for n, p in enc.named_parameters()), reverse=True)
    _log(f"[vae3d encoder backward {str(dtype)[6:]}] forward rel {rel(ya, yr):.2e}; dL/dx rel {e_x:.2e}; parameters: worst {errs[0][0]:.2e} "
         f"({errs[0][1]}), median {errs[len(errs) // 2][0]:.2e} ({len(names)} tensors)")
    assert e_x <= NET_IN_TOL[dtype], e_x
    assert errs[0][0] <= NET_W_TOL[dtype], errs[:5]


@pytest.mark.parametrize("dtype", DT)
def test_vae3d_decoder_backward_vs_autograd_of_the_oracle(dtype):
    """the SD2.1-compatible family's Decoder (vae_models.py:960-1002): mid block with the spatial-temporal attention block
    (LayerNorm + per-pixel attention over time, one outer residual), Upsample3D with zero H / W padding in both kinds"""
    import cvvae_amd
    over = dict(ch=128, ch_mult=(1, 2, 4), num_res_blocks=1)
    m = cvvae_amd.CVVAEModel(**over)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 10)
    m.load_state_dict(sd, strict=True)
    m = m.to(dtype).cuda()
    dec = m.decoder.train()
    ref_sd = {k: v.to(dtype).float().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("decoder.")}
    z = seeded_input((1, 4, 3, 8, 8), 15).to(dtype)
    zr = z.float().clone().requires_grad_(True)
    yr = O.v3_decoder(zr, ref_sd, dict(over))
    cot = seeded_input(tuple(yr.shape), 7).to(dtype)
    (yr * cot.float()).sum().backward()
    za = z.cuda().requires_grad_(True)
    ya = dec(za)
    (ya.float() * cot.cuda().float()).sum().backward()
    e_z = rel(za.grad, zr.grad)
    names = [n for n, _ in dec.named_parameters()]
    scale = max(float(ref_sd["decoder." + n].grad.norm()) for n in names)
    errs = sorted(((rel(p.grad, ref_sd["decoder." + n].grad, 1e-3 * scale), n) for n, p in dec.named_parameters()), reverse=True)
    _log(f"[vae3d decoder backward {str(dtype)[6:]}] forward rel {rel(ya, yr):.2e}; dL/dz rel {e_z:.2e}; parameters: worst {errs[0][0]:.2e} "
         f"({errs[0][1]}), median {errs[len(errs) // 2][0]:.2e} ({len(names)} tensors)")
    assert e_z <= NET_IN_TOL[dtype], e_z
    assert errs[0][0] <= NET_W_TOL[dtype], errs[:5]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_training_step_chain_through_the_frozen_constraint_decoder(dtype):
    """the reference's step (autoencoder.py:1057-1069): z from the TRAINABLE encoder, xrec_2d = frozen constraint_decoder(z), a loss
    on xrec_2d; .backward() must fill every encoder parameter's .grad -- compared with autograd over the oracle's two networks."""
    from cvvae_amd.constraint import DecoderWith3DWrapper
    from oracle.golden_cases import CONSTRAINT_CFG
    cfg = dict(CONSTRAINT_CFG, up_block_types=["UpDecoderBlock2D"] * 2, block_out_channels=[128, 256], layers_per_block=1)
    m, ref_sd = _encoder(dtype)
    enc = m.encoder.train()
    dec = DecoderWith3DWrapper(**cfg)
    dsd = seeded_state_dict({k: v.shape for k, v in dec.state_dict().items()}, 9)
    dec.load_state_dict(dsd, strict=True)
    dec = dec.to(dtype).cuda().eval().requires_grad_(False)
    dref = {k: v.to(dtype).float() for k, v in dsd.items()}
    x = seeded_input((1, 3, 5, 32, 32), 12).to(dtype)
    zr = O.sd3_encoder(x.float(), ref_sd, dict(SMALL))[:, :16]
    yr = O.constraint_decoder(zr, dref, cfg)
    cot = seeded_input(tuple(yr.shape), 6).to(dtype)
    (yr * cot.float()).sum().backward()
    z = enc(x.cuda())[:, :16].contiguous()
    y = dec(z)
    (y.float() * cot.cuda().float()).sum().backward()
    names = [n for n, _ in enc.named_parameters()]
    scale = max(float(ref_sd["encoder." + n].grad.norm()) for n in names)
    errs = sorted(((rel(p.grad, ref_sd["encoder." + n].grad, 1e-3 * scale), n) for n, p in enc.named_parameters()), reverse=True)
    _log(f"[training chain {str(dtype)[6:]}] loss(constraint_decoder(encoder(x))): xrec rel {rel(y, yr):.2e}; encoder parameter "
         f"gradients worst {errs[0][0]:.2e} ({errs[0][1]}), median {errs[len(errs) // 2][0]:.2e}")
    assert errs[0][0] <= 2 * NET_W_TOL[dtype], errs[:5]


def test_last_layer_gradient_probe_before_the_backward():
    """torch.autograd.grad(loss, decoder.get_last_layer(), retain_graph=True) -- the reference's adaptive adversarial weight,
    lvdm/modules/autoencoding/losses/discriminator_loss.py:211-220 -- runs the tail node alone (GroupNorm + SiLU + conv_out), twice,
    then the step's real backward: the probe equals the conv_out.weight gradient of a full backward on the same cotangent, bit for
    bit (same launches), and the full backward after the probes still fills every parameter"""
    import cvvae_amd
    from cvvae_amd import ops
    dtype = torch.bfloat16
    m = cvvae_amd.CVVAESD3Model(**SMALL)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 8)
    m.load_state_dict(sd, strict=True)
    dec = m.decoder.to(dtype).cuda().train()
    z = seeded_input((1, 16, 3, 8, 12), 13).to(dtype).cuda()
    calls = []
    real = ops.conv_wgrad
    ops.conv_wgrad = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        y = dec(z.clone().requires_grad_(True))
        cot = seeded_input(tuple(y.shape), 5).cuda().to(y.dtype)
        loss = (y.float() * cot.float()).sum()
        g1 = torch.autograd.grad(loss, dec.get_last_layer(), retain_graph=True)[0]
        assert len(calls) == 1, len(calls)
        g2 = torch.autograd.grad(loss, dec.get_last_layer(), retain_graph=True)[0]
        assert torch.equal(g1, g2)
        loss.backward()
    finally:
        ops.conv_wgrad = real
    assert torch.equal(dec.conv_out.weight.grad, g1)
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in dec.parameters())
    _log(f"[last-layer probe] conv_out.weight gradient alone: 1 weight-gradient launch (a full backward: {len(calls) - 2}); "
        f"probe == full backward's conv_out.weight.grad bit for bit")


def test_autocast_training_step_on_fp32_masters():
    """torch.autocast(dtype=bfloat16) over the fp32 model (precision bf16-mixed of the reference's trainer, main.py:905-912): 16-bit
    launches on copies of the weights, fp32 gradients for the masters; equal to a model that holds the bf16 weights (forward and
    input gradient bit for bit, parameter gradients after rounding), and ~2.5x faster than the fp32 model's split-precision step"""
    import copy

    import cvvae_amd
    m = cvvae_amd.CVVAESD3Model(**SMALL)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 7)
    m.load_state_dict(sd, strict=True)
    m16 = copy.deepcopy(m)
    for p in m16.parameters():
        if p.dim() >= 2:
            p.data = p.data.to(torch.bfloat16)
    m, m16 = m.cuda().train(), m16.cuda().train()
    x = seeded_input((1, 3, 5, 32, 32), 11).cuda()

    def step(model, xin, autocast):
        xin = xin.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            mom = model.encoder(xin)
            rec = model.decoder(mom[:, :16].contiguous())
            loss = (rec.float() - x.float()).pow(2).mean() + 1e-3 * mom.float().pow(2).mean()
        probe = torch.autograd.grad(loss, model.decoder.get_last_layer(), retain_graph=True)[0]
        loss.backward()
        return mom, rec, xin.grad, probe

    mom, rec, gx, probe = step(m, x, True)
    mom16, rec16, gx16, probe16 = step(m16, x.to(torch.bfloat16), False)
    assert mom.dtype == torch.bfloat16 and torch.equal(mom, mom16) and torch.equal(rec, rec16)
    assert gx.dtype == torch.float32 and torch.equal(gx.to(torch.bfloat16), gx16)
    assert probe.dtype == torch.float32 and torch.equal(probe.to(torch.bfloat16), probe16)
    for (n, p), (_, q) in zip(m.named_parameters(), m16.named_parameters()):
        if n.startswith("quant") or n.startswith("post_quant"):
            continue
        assert p.dtype == torch.float32 and p.grad is not None and p.grad.dtype == torch.float32, n
        assert torch.equal(p.grad.to(q.grad.dtype), q.grad), n
    _log("[autocast bf16 over fp32 masters] forward, dL/dx and the last-layer probe equal the bf16-weight model's bit for bit; "
         f"{sum(1 for p in m.parameters() if p.grad is not None)} fp32 parameter gradients equal after rounding")
