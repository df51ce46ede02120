"""Input gradients through the frozen 2-D constraint decoder (cvvae_amd/grad.py; lvdm/models/autoencoder.py:1057-1069: the
latent-compatibility loss back-propagates through `constraint_decoder(z)` into the latents).  Every new kernel, every block's
backward and the whole decoder are compared with torch autograd over the ORACLE (plain PyTorch fp32 on the CPU, same dtype-rounded
weights and inputs).  Errors are relative L2 norms of the gradient; tolerances: fp32 models 2e-3, fp16 2e-2, bf16 1e-1 for the whole
31-layer decoder (printed values are recorded in DESIGN.md), tighter for single ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cvvae_oracle as O
from oracle.golden_cases import CONSTRAINT_CASES, CONSTRAINT_CFG
from oracle.seeded import seeded_input, seeded_state_dict

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.float16, torch.bfloat16]
OP_TOL = {torch.float32: 2e-5, torch.float16: 3e-3, torch.bfloat16: 2e-2}      # one op: output rounding of the storage dtype
BLOCK_TOL = {torch.float32: 3e-4, torch.float16: 8e-3, torch.bfloat16: 5e-2}
NET_TOL = {torch.float32: 2e-3, torch.float16: 2e-2, torch.bfloat16: 1e-1}


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def decoder(dtype, wseed=0):
    from cvvae_amd.constraint import DecoderWith3DWrapper
    m = DecoderWith3DWrapper(**CONSTRAINT_CFG)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, wseed)
    m.load_state_dict(sd, strict=True)
    m = m.to(dtype).cuda().eval().requires_grad_(False)  # lvdm/models/autoencoder.py:1057-1058
    return m, {k: v.to(dtype).float() for k, v in sd.items()}  # the oracle sees the same rounded weights


def nchw(x):  # [N,1,H,W,C] device tensor -> fp32 CPU [N,C,H,W]
    return x.detach().float().cpu()[:, 0].permute(0, 3, 1, 2).contiguous()


def ndhwc(x, dtype):  # fp32 CPU [N,C,H,W] -> [N,1,H,W,C] device tensor
    return x.permute(0, 2, 3, 1).unsqueeze(1).contiguous().to(dtype).cuda()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C,H,W,per_frame,silu,with_add", [(128, 80, 80, False, True, True), (256, 24, 16, True, True, False),
                                                          (512, 9, 7, False, False, True), (128, 8, 8, True, False, False)])
def test_gn_bwd_input(dtype, C, H, W, per_frame, silu, with_add):
    """cvvae_gn_bwd_input vs autograd of act(F.group_norm(x)) (several pixel splits at 80x80; strongly offset activations)"""
    from cvvae_amd import ops
    torch.manual_seed(C + H)
    B, T = 2, 3
    x = (torch.randn(B, T, H, W, C) * 1.5 + 2.0).to(dtype)
    gy = torch.randn(B, T, H, W, C).to(dtype)
    add = torch.randn(B, T, H, W, C).to(dtype) if with_add else None
    gamma, beta = torch.randn(C) * 0.5 + 1.0, torch.randn(C) * 0.3
    xr = x.float().requires_grad_(True)
    if per_frame:
        f = xr.permute(0, 1, 4, 2, 3).reshape(B * T, C, H * W)
    else:
        f = xr.permute(0, 4, 1, 2, 3).reshape(B, C, T * H * W)
    y = F.group_norm(f, 32, gamma, beta, 1e-6)
    y = F.silu(y) if silu else y
    gyf = gy.float().permute(0, 1, 4, 2, 3).reshape(B * T, C, H * W) if per_frame else gy.float().permute(0, 4, 1, 2, 3).reshape(B, C, -1)
    y.backward(gyf)
    ref = xr.grad + (add.float() if with_add else 0.0)
    xd = x.cuda()
    one, zero = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    tabs = ops.gn_stats(xd, one, zero, 1e-6, per_frame=per_frame)
    got = ops.gn_bwd_input(xd, gy.cuda(), tabs, gamma.cuda(), beta.cuda(), silu, add=add.cuda() if with_add else None,
                           per_frame=per_frame)
    e = rel(got, ref)
    print(f"\n[gn_bwd {str(dtype)[6:]} C{C} {H}x{W} per_frame={per_frame} silu={silu}] rel {e:.2e}")
    assert e <= OP_TOL[dtype]
    assert torch.equal(got, ops.gn_bwd_input(xd, gy.cuda(), tabs, gamma.cuda(), beta.cuda(), silu,
                                             add=add.cuda() if with_add else None, per_frame=per_frame)), "not deterministic"


@pytest.mark.parametrize("dtype", DT)
def test_softmax_bwd_upsample_sum_transpose(dtype):
    from cvvae_amd import ops
    torch.manual_seed(3)
    rows, n, ld = 96, 80, 128
    s = torch.randn(rows, n) * 2
    p = torch.softmax(s, -1)
    pd = torch.zeros(rows, ld)
    pd[:, :n] = p
    pd = pd.to(dtype)
    gp = torch.zeros(rows, ld)
    gp[:, :n] = torch.randn(rows, n)
    pr = pd.float()[:, :n]
    ref = 0.25 * pr * (gp[:, :n] - (pr * gp[:, :n]).sum(-1, keepdim=True))
    got = ops.softmax_bwd_rows(pd.cuda(), gp.cuda(), n, 0.25)
    assert rel(got[:, :n], ref) <= OP_TOL[dtype] and float(got[:, n:].float().abs().max()) == 0.0
    g = torch.randn(3, 1, 10, 12, 64).to(dtype)
    ref = g.float().view(3, 1, 5, 2, 6, 2, 64).sum((3, 5))
    assert rel(ops.upsample2x_sum(g.cuda()), ref) <= OP_TOL[dtype]
    x = torch.randn(2, 40, 128).to(dtype)
    t = ops.transpose(x.cuda(), ncols=40, ld_out=128)
    assert tuple(t.shape) == (2, 40, 128) and torch.equal(t[:, :, :40].cpu(), x[:, :, :40].transpose(1, 2))
    assert float(t[:, :, 40:].float().abs().max()) == 0.0
    assert torch.equal(ops.transpose(x.cuda()).cpu(), x.transpose(1, 2))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("pre,cin,cout", [("up_blocks.3.resnets.0.conv1", 256, 128), ("conv_out", 128, 3), ("conv_in", 16, 512),
                                          ("up_blocks.2.resnets.0.conv_shortcut", 512, 256)])
def test_conv_input_gradient(dtype, pre, cin, cout):
    """conv(gy, WeightCache.conv_dgrad(...)) vs autograd's grad_input of F.conv2d(padding=1) / the 1x1 shortcut"""
    from cvvae_amd import _lib as L, grad, ops
    from cvvae_amd.engine import P2D, ZERO
    m, sd = decoder(dtype)
    wc = m._cache()
    N, H, W = 2, 12, 20
    w = sd[pre + ".weight"]
    assert w.shape[0] == cout and w.shape[1] == cin
    x = torch.randn(N, cin, H, W, requires_grad=True)
    gy = torch.randn(N, cout, H, W).to(dtype).float()
    F.conv2d(x, w, None, padding=w.shape[-1] // 2).backward(gy)
    if w.shape[-1] == 1:
        got = nchw(grad._dgrad1x1(wc, ndhwc(gy, dtype), pre))
    elif cout == 3:  # channel-padded cotangent, as the decoder's backward feeds it
        g = torch.zeros(N, 1, H, W, 32, dtype=dtype, device="cuda")
        g[..., :3] = ndhwc(gy, dtype)
        got = nchw(grad._dgrad3x3(wc, g, pre, cin_pad=32))
    elif cin == 16:  # NCDHW store of the narrow latent gradient
        got = ops.conv(ndhwc(gy, dtype), wc.conv_dgrad(pre, (1, 3, 3)), pad=P2D, pad_mode_hw=ZERO, out_mode=L.OUT_NCDHW)[:, :, 0].float().cpu()
    else:
        got = nchw(grad._dgrad3x3(wc, ndhwc(gy, dtype), pre))
    e = rel(got, x.grad)
    print(f"\n[dgrad {pre} {str(dtype)[6:]}] rel {e:.2e}")
    assert e <= OP_TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("pre", ["mid_block.resnets.0", "up_blocks.2.resnets.0"])
def test_resnet_backward(dtype, pre):
    from cvvae_amd import engine, grad
    m, sd = decoder(dtype)
    wc = m._cache()
    cin = sd[pre + ".conv1.weight"].shape[1]
    cout = sd[pre + ".conv1.weight"].shape[0]
    N, H, W = 2, 16, 12
    x = (torch.randn(N, cin, H, W) * 0.8).to(dtype).float().requires_grad_(True)
    gy = torch.randn(N, cout, H, W).to(dtype).float()
    yr = O.c2d_resnet(x, sd, pre)
    yr.backward(gy)
    tape = []
    y, _ = engine.c2d_resnet(wc, ndhwc(x.detach(), dtype), None, pre, tape=tape)
    assert rel(nchw(y), yr) <= BLOCK_TOL[dtype]
    got = grad.resnet_backward(wc, ndhwc(gy, dtype), tape[0])
    e = rel(nchw(got), x.grad)
    print(f"\n[resnet_backward {pre} {str(dtype)[6:]}] rel {e:.2e}")
    assert e <= BLOCK_TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("H,W", [(8, 8), (12, 8)])
def test_attention_backward(dtype, H, W):
    """64 and 96 tokens (the second pads the score rows to 128 columns)"""
    from cvvae_amd import engine, grad
    m, sd = decoder(dtype)
    wc = m._cache()
    a = "mid_block.attentions.0"
    N, C = 3, 512
    x = (torch.randn(N, C, H, W) * 0.7).to(dtype).float().requires_grad_(True)
    gy = torch.randn(N, C, H, W).to(dtype).float()
    yr = O.sd3_attention(x.unsqueeze(2), sd, a).squeeze(2)
    yr.backward(gy)
    tape = []
    y = engine.spatial_attention(wc, ndhwc(x.detach(), dtype), a + ".group_norm", a + ".to_q", a + ".to_k", a + ".to_v",
                                 a + ".to_out.0", 1e-6, True, tape=tape)
    assert rel(nchw(y), yr) <= BLOCK_TOL[dtype]
    got = grad.attention_backward(wc, ndhwc(gy, dtype), tape[0])
    e = rel(nchw(got), x.grad)
    print(f"\n[attention_backward {H}x{W} {str(dtype)[6:]}] rel {e:.2e}")
    assert e <= BLOCK_TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("name", ["constraint2d_t3_8", "constraint2d_4d_12x8"])
def test_constraint_decoder_input_gradient(name, dtype):
    """the whole frozen decoder under torch.autograd: z.grad of a fixed linear functional of the reconstruction vs the oracle's"""
    cfg, zshape, wseed, zseed = CONSTRAINT_CASES[name]
    m, sd = decoder(dtype, wseed)
    z0 = seeded_input(zshape, zseed).to(dtype)
    zr = z0.float().clone().requires_grad_(True)  # (a copy: .float() of an fp32 tensor is the tensor itself)
    yr = O.constraint_decoder(zr, sd, cfg)
    cot = seeded_input(tuple(yr.shape), 77)
    (yr * cot).sum().backward()
    z = z0.cuda().requires_grad_(True)
    y = m(z)
    assert y.requires_grad and y.dtype == dtype
    with torch.no_grad():
        assert torch.equal(y, m(z0.cuda())), "the taped forward must be the inference forward"
    (y.float() * cot.cuda()).sum().backward()
    assert z.grad is not None and z.grad.dtype == dtype and z.grad.shape == z.shape
    e = rel(z.grad, zr.grad)
    cos = float(F.cosine_similarity(z.grad.float().cpu().flatten(), zr.grad.flatten(), dim=0))
    print(f"\n[constraint decoder input gradient {name} {str(dtype)[6:]}] rel L2 {e:.2e} cosine {cos:.6f} |g| {float(zr.grad.norm()):.3e}")
    assert e <= NET_TOL[dtype] and cos >= 1.0 - NET_TOL[dtype]
    # a second backward through a fresh forward gives the same bits (deterministic kernels)
    z2 = z0.cuda().requires_grad_(True)
    (m(z2).float() * cot.cuda()).sum().backward()
    assert torch.equal(z2.grad, z.grad)


def test_constraint_decoder_refuses_weight_gradients():
    m, _ = decoder(torch.float16)
    m.conv_in.weight.requires_grad_(True)
    z = torch.zeros(1, 16, 1, 8, 8, dtype=torch.float16, device="cuda", requires_grad=True)
    with pytest.raises(NotImplementedError):
        m(z)
