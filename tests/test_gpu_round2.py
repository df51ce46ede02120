"""Round-2 GPU parity cases: forward() on the device, the latent pre-compute API, op-level spatial attention at 4096 tokens
against fp32 SDPA, the fused GroupNorm statistics on strongly offset activations (|mean| >= 20 sigma) against fp64 moments,
temporal attention beyond 8 frames, the u8 pre-processing through tiles, a non-current device, and the RCCL window shard on
two GPUs (skipped on a 1-GPU box)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cvvae_oracle as O
from oracle.seeded import seeded_input, seeded_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"
REP, ZERO = 1, 0


def build(family, over, dtype, wseed, device="cuda"):
    import cvvae_amd
    cls = cvvae_amd.CVVAESD3Model if family == "sd3" else cvvae_amd.CVVAEModel
    m = cls(**over)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, wseed)
    m.load_state_dict(sd, strict=True)
    return m.to(dtype).to(device).eval(), sd


@pytest.mark.parametrize("family", ["sd3", "vae3d"])
def test_forward_on_device(family):
    """CVVAE*Model.forward (modeling_vae.py:114-142 / 440-468): sample_posterior with a generator (device and CPU generators),
    num_frames on 4-D input, return_dict=False -- against the explicit encode / sample / decode sequence, bit for bit."""
    dtype = torch.float16
    m, _ = build(family, {}, dtype, 4)
    x = seeded_input((1, 3, 5, 64, 64), 31).to(dtype).cuda()
    post = m.encode(x).latent_dist
    out = m(x)
    assert torch.equal(out.sample, m.decode(post.mode()).sample) and torch.equal(out[0], out.sample)
    assert torch.equal(m(x, return_dict=False)[0], out.sample)
    for gdev in ("cuda", "cpu"):
        z = post.sample(generator=torch.Generator(device=gdev).manual_seed(7))
        a = m(x, sample_posterior=True, generator=torch.Generator(device=gdev).manual_seed(7)).sample
        b = m(x, sample_posterior=True, generator=torch.Generator(device=gdev).manual_seed(7)).sample
        assert torch.equal(a, b) and torch.equal(a, m.decode(z).sample)
    assert not torch.equal(a, out.sample)
    # 4-D frames in, num_frames for the decode (T2I-style plumbing): same numbers as the 5-D call
    m4, _ = build(family, {"num_video_frames": 5}, dtype, 4)
    x4 = x.permute(0, 2, 1, 3, 4).reshape(5, 3, 64, 64)
    assert torch.equal(m4.encode(x4).latent_dist.mode(), post.mode())
    z5 = post.mode()
    z4 = z5.permute(0, 2, 1, 3, 4).reshape(-1, z5.shape[1], *z5.shape[3:])
    assert torch.equal(m4.decode(z4, num_frames=z5.shape[2]).sample, out.sample)
    assert torch.equal(m4(x4, num_frames=z5.shape[2]).sample, out.sample)


def test_encode_latents_precompute():
    """the frozen-encoder latent pre-compute (cfg 5 generalised; lvdm/models/diffusion.py:159-171): a batch of clips in rounds,
    mode or seeded samples, scaled -- equal to per-clip encode calls."""
    dtype = torch.bfloat16
    m, sd = build("sd3", {}, dtype, 1)
    x = seeded_input((3, 3, 9, 64, 96), 12).to(dtype).cuda()
    z = m.encode_latents(x, sample=False)
    want = torch.cat([m.encode(x[i:i + 1]).latent_dist.mode() for i in range(3)], dim=0)
    assert torch.equal(z, want) and z.shape == (3, 16, 3, 8, 12)
    assert torch.equal(m.encode_latents(x, sample=False, n_samples_a_time=2), z)
    assert torch.equal(m.encode_latents(x, sample=False, scale_factor=0.5), 0.5 * want)
    a = m.encode_latents(x, generator=torch.Generator(device="cuda").manual_seed(3))
    assert torch.equal(a, m.encode_latents(x, generator=torch.Generator(device="cuda").manual_seed(3)))
    with torch.no_grad():
        ref = O.posterior_mode(O.encode_moments(x[:1].float().cpu(), sd, {}, "sd3"))
    assert (z[:1].float().cpu() - ref).abs().max() <= 3.5e-2
    mv, _ = build("vae3d", {}, dtype, 1)
    zi = mv.encode_latents(x[:, :, 0], sample=False)  # images -> [(B T'), z, h, w], scaled by config.scaling_factor
    assert zi.shape == (3, 4, 8, 12)
    assert torch.equal(zi, 0.18215 * mv.encode(x[:, :, :1]).latent_dist.mode()[:, :, 0])


def _attn_module(C):
    from cvvae_amd.modeling import ConvP, NormP, _holder
    torch.manual_seed(0)
    a = _holder(group_norm=NormP(C), to_q=ConvP(C, C, ()), to_k=ConvP(C, C, ()), to_v=ConvP(C, C, ()), to_out=ConvP(C, C, ()))
    with torch.no_grad():
        a.group_norm.weight.normal_(1.0, 0.1)
        a.group_norm.bias.normal_(0.0, 0.1)
    return a


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("hw", [(64, 64), (36, 40)])
def test_spatial_attention_4096_tokens_vs_fp32_sdpa(dtype, hw):
    """AttentionWithExtraDim / diffusers Attention (vae_blocks3d_sd3.py:119-147) at BASELINE cfg 3's size:
    2 frames x 4096 tokens x 512 channels (and a token count that is NOT a multiple of the 128-token padding: 1440), op level,
    against GroupNorm -> Linear q/k/v -> F.scaled_dot_product_attention -> Linear -> + residual in fp32 on the same rounded
    inputs and weights."""
    from cvvae_amd import engine
    C = 512
    H, W = hw
    a = _attn_module(C).to(dtype).cuda()
    wc = engine.WeightCache(a)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn((1, 2, H, W, C), generator=g) * 1.5 + 0.3).to(dtype).cuda()
    y = engine.spatial_attention(wc, x, "group_norm", "to_q", "to_k", "to_v", "to_out", 1e-6, True)
    assert y.shape == x.shape
    # fp32 reference on the device (torch ops are the yardstick here, never the product path)
    xf = x.float()
    W32 = {k: v.float() for k, v in a.state_dict().items()}
    ref = []
    for t in range(2):
        xt = xf[0, t].reshape(H * W, C)                                    # tokens x C
        n = F.group_norm(xt.t().unsqueeze(0), 32, W32["group_norm.weight"], W32["group_norm.bias"], 1e-6)[0].t()
        q = F.linear(n, W32["to_q.weight"], W32["to_q.bias"])
        k = F.linear(n, W32["to_k.weight"], W32["to_k.bias"])
        v = F.linear(n, W32["to_v.weight"], W32["to_v.bias"])
        o = F.scaled_dot_product_attention(q[None, None], k[None, None], v[None, None])[0, 0]
        ref.append(F.linear(o, W32["to_out.weight"], W32["to_out.bias"]) + xt)
    ref = torch.stack(ref).reshape(1, 2, H, W, C)
    err = (y.float() - ref).abs()
    scale = ref.abs().max().item()
    # storage rounding of q, k, v, P and the output: a few ulps of the dtype relative to the data range
    tol = (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * 3
    print(f"\n[attention {H*W} tokens {str(dtype)[6:]}] max|d| {err.max().item():.3e} mean|d| {err.mean().item():.3e} range {scale:.2f}")
    assert err.max().item() <= tol * scale and err.mean().item() <= tol * scale * 0.1


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_gn_stats_large_mean_vs_fp64(dtype):
    """Numerics of the fused GroupNorm statistics (shifted sums in the conv epilogue): activations whose group means sit 20-60
    sigma away from zero.  The (scale, shift) tables of cvvae_conv_fwd_gn + cvvae_gn_finalize are compared with fp64 moments of
    the very tensor the conv stored; fast tail (full tiles) and general tail (ragged tile edges) both."""
    from cvvae_amd import ops
    for shape, Cout in (((1, 4, 32, 64), 128), ((2, 3, 19, 37), 256)):
        B, T, H, W = shape
        g = torch.Generator().manual_seed(3)
        x = torch.randn((B, T, H, W, 128), generator=g).to(dtype).cuda()
        w = (torch.randn((Cout, 128, 1, 3, 3), generator=g) / (128 * 9) ** 0.5).to(dtype).cuda()
        bias = (torch.randn(Cout, generator=g).sign() * (20.0 + 40.0 * torch.rand(Cout, generator=g))).cuda()  # |mean| = 20..60 sigma
        pw = ops.pack_weight(w.reshape(Cout, 128, 9), bias, (1, 3, 3))
        y, part = ops.conv(x, pw, pad=((0, 0), (1, 1), (1, 1)), gn_out=32)
        gamma = torch.ones(Cout, device=DEV)
        beta = torch.zeros(Cout, device=DEV)
        sc, sh = ops.gn_finalize(part, gamma, beta, 1e-6)
        y64 = y.double().reshape(B, -1, 32, Cout // 32)
        mean = y64.mean(dim=(1, 3))
        var = y64.var(dim=(1, 3), unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-6)
        sc_ref = rstd.repeat_interleave(Cout // 32, dim=1)
        sh_ref = (-mean * rstd).repeat_interleave(Cout // 32, dim=1)
        # sigma of a group also carries the spread of its channels' biases; what matters is the relative error of rstd
        rel = ((sc.double() - sc_ref) / sc_ref).abs().max().item()
        # normalised values (x*scale+shift) are O(1): the shift must be right to ~1e-4 of them although |shift| ~ 20-60
        esh = (sh.double() - sh_ref).abs().max().item()
        print(f"\n[gn large mean {str(dtype)[6:]} {shape}] rstd rel err {rel:.2e}, shift abs err {esh:.2e} (|shift| up to {sh_ref.abs().max().item():.1f})")
        assert rel <= 2e-5 and esh <= 5e-4


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T", [9, 20])
def test_temporal_attention_more_than_8_frames(dtype, T):
    from cvvae_amd import ops
    g = torch.Generator().manual_seed(T)
    q, k, v = [(torch.randn((2, T, 3, 7, 512), generator=g)).to(dtype) for _ in range(3)]
    o = ops.temporal_attention(q.cuda(), k.cuda(), v.cuda()).float().cpu()
    tok = lambda t: t.float().permute(0, 2, 3, 1, 4).reshape(-1, T, 512)  # noqa: E731
    sc = torch.softmax(tok(q) @ tok(k).transpose(1, 2) * 512 ** -0.5, -1)
    oref = (sc @ tok(v)).reshape(2, 3, 7, T, 512).permute(0, 3, 1, 2, 4)
    tol = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert (o - oref).abs().max() <= tol * oref.abs().max() * 1.5


def test_vae3d_decode_without_temporal_chunking():
    """en_de_n_frames_a_time=None (one network call over the whole clip): the vae3d decoder's temporal attention then sees 9
    latent frames (> 8: the general kernel).  Against the CPU oracle with the same configuration."""
    dtype = torch.float16
    cfg = {"en_de_n_frames_a_time": None}
    m, sd = build("vae3d", cfg, dtype, 6)
    g = torch.Generator().manual_seed(9)
    z = torch.randn((1, 4, 9, 8, 8), generator=g)
    with torch.no_grad():
        ref = O.decode_sample(z.to(dtype).float(), sd, cfg, "vae3d")
    y = m.decode(z.to(dtype).cuda()).sample
    assert y.shape == (1, 3, 33, 64, 64)
    assert (y.float().cpu() - ref).abs().max() <= 1.5e-2


def test_u8_preprocessing_through_tiles_and_image_mode():
    """encode_frames_u8 feeds the padded NDHWC clip to the encoder directly (no NCDHW round trip): spatial tiles and temporal
    windows are cut from the NDHWC clip.  Must equal encode() of the script-normalised NCDHW clip bit for bit."""
    dtype = torch.float16
    m, _ = build("sd3", {"tile_spatial_size": 144}, dtype, 0)
    g = torch.Generator().manual_seed(2)
    for T in (21, 1):
        u8 = torch.randint(0, 256, (T, 160, 200, 3), generator=g, dtype=torch.uint8)
        frame_end = 1 + (T - 1) // 4 * 4
        video = (u8.permute(3, 0, 1, 2).unsqueeze(0).to(dtype) / 127.5 - 1.0)[:, :, :frame_end].cuda()
        want = m.encode(video).latent_dist.parameters
        got = m.encode_frames_u8(u8.cuda()).latent_dist.parameters
        assert torch.equal(got, want), f"T={T}"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_model_on_non_current_device():
    """model and input on cuda:1 while the current device is cuda:0 (plain PyTorch, i.e. the reference, handles this): launches
    must go to cuda:1's stream; result equals the cuda:0 run bit for bit."""
    dtype = torch.bfloat16
    m0, _ = build("sd3", {"tile_spatial_size": 144}, dtype, 0, "cuda:0")
    m1, _ = build("sd3", {"tile_spatial_size": 144}, dtype, 0, "cuda:1")
    x = seeded_input((1, 3, 5, 160, 200), 1).to(dtype)
    torch.cuda.set_device(0)
    z0 = m0.encode(x.to("cuda:0")).latent_dist.parameters
    z1 = m1.encode(x.to("cuda:1")).latent_dist.parameters
    assert z1.device.index == 1 and torch.equal(z0.cpu(), z1.cpu())
    y1 = m1.decode(z1[:, :16]).sample
    assert torch.equal(m0.decode(z0[:, :16]).sample.cpu(), y1.cpu())
    from cvvae_amd import ops
    with pytest.raises(RuntimeError, match="current device"):
        ops.transpose(torch.zeros((1, 8, 8), dtype=dtype, device="cuda:1"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _nccl_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from cvvae_amd import dist as D
        dtype = torch.bfloat16
        m, _ = build("sd3", {"tile_spatial_size": 144}, dtype, 0, f"cuda:{rank}")
        T = 49
        x = seeded_input((1, 3, T, 160, 200), 3).to(dtype).cuda()
        full = m.encode(x).latent_dist.parameters
        a, b = D.owned_frames(T, 16, world, rank)
        got = D.encode_windows_sharded(m, x[:, :, a:b].contiguous(), T_total=T, time_sharded=True)  # isend/irecv + all_gather
        ok_e = torch.equal(got, full)
        z = full[:, :16].contiguous()
        Tz = z.shape[2]
        a, b = D.owned_frames(Tz, 4, world, rank)
        y = D.decode_windows_sharded(m, z[:, :, a:b].contiguous(), T_total=Tz, time_sharded=True)
        ok_d = torch.equal(y, m.decode(z).sample)
        # (window x tile) units: a ONE-window clip whose 2x2 tiles are split over the two ranks (raw tiles by batched send/recv)
        x1 = x[:, :, :17].contiguous()
        ok_e = ok_e and torch.equal(D.encode_units_sharded(m, x1), m.encode(x1).latent_dist.parameters)
        z1 = z[:, :, :5].contiguous()
        ok_d = ok_d and torch.equal(D.decode_units_sharded(m, z1), m.decode(z1).sample)
        # ... and bench.py's own N > 1 step (what `bench.py --gpus 2` times): one clip of 33 frames, time-sharded, checked per rank
        import bench
        Tb = bench.temporal_shard_T(world)
        xf, xl = bench.temporal_shard_input(Tb, 96, 128, world, rank, dtype, f"cuda:{rank}")
        mom, yl = bench.temporal_shard_step(m, xl, Tb)
        ok_e = ok_e and bench.temporal_shard_check(m, xf, mom, yl, world, rank, f"cuda:{rank}")
        q.put((rank, ok_e, ok_d))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL window shard)")
def test_window_shard_over_rccl_two_gpus():
    """cvvae_amd/dist.py on the real transport: two processes, one per GPU, time-sharded clip, boundary frame by RCCL
    send/recv, latents all-gathered -- equal to the single-process wrapper bit for bit."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok_e and ok_d for _, ok_e, ok_d in res), res


def L_desc_name(ops, dtype):
    """kernel instance the library would pick for a per-frame 128-channel conv with prologue (under the current CVVAE_CONV_FORCE)"""
    from cvvae_amd import _lib as L
    d = L.ConvDesc()
    d.dtype = ops._dt(dtype)
    d.B, d.Ti, d.Hi, d.Wi, d.Cin, d.in_pix_stride = 2, 5, 40, 72, 128, 128
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW = 1, 3, 3, 1, 1, 1
    d.pad_t, d.pad_h, d.pad_w, d.prologue, d.gn_rows_per_batch = 0, 1, 1, 1, 1
    d.To, d.Ho, d.Wo, d.Cout, d.out_pix_stride, d.alpha = 5, 40, 72, 128, 128, 1.0
    return ops.conv_kernel_name(d)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_four_wave_instances_agree_bit_for_bit(dtype, monkeypatch):
    """The 4-wave instances (two resident workgroups per CU) accumulate every output in the same order as the 8-wave ones: the
    same launch under either tiling must agree BIT FOR BIT -- per-frame 3x3 with GN+SiLU prologue, residual and fused
    statistics; causal 3x3x3 with time folds over an odd frame count (two-frame tiles whose frames have different fold plans
    run the plain three-group walk, the last frame goes to the one-frame sibling)."""
    from cvvae_amd import ops
    monkeypatch.setenv("CVVAE_CONV_FORCE", "1x8x32:1x4x1:2")
    probe = L_desc_name(ops, dtype)
    monkeypatch.delenv("CVVAE_CONV_FORCE")
    if probe is None or "w1x4x1" not in probe:
        pytest.skip("the four-wave per-frame instance is not in this build")
    g = torch.Generator().manual_seed(11)
    x = torch.randn((2, 5, 40, 72, 128), generator=g).to(dtype).cuda()
    res = torch.randn((2, 5, 40, 72, 128), generator=g).to(dtype).cuda()
    gsc = (1.0 + 0.1 * torch.randn((2, 128), generator=g)).cuda()
    gsh = (0.1 * torch.randn((2, 128), generator=g)).cuda()
    names = []
    ops.PROFILE = lambda d, pw_, launch: (names.append(ops.conv_kernel_name(d)), launch())
    try:
        w2 = (torch.randn((128, 128, 1, 3, 3), generator=g) / (128 * 9) ** 0.5).to(dtype).cuda()
        pw2 = ops.pack_weight(w2.reshape(128, 128, 9), torch.randn(128, generator=g).cuda(), (1, 3, 3))
        for with_res in (False, True):
            # (the residual is pre-accumulated during the K loop at a chunk that depends on the fragments per wave: with it the
            #  two tilings round differently in the last bit, without it they are bit-identical)
            kw = dict(pad=((0, 0), (1, 1), (1, 1)), prologue=1, gn=(gsc, gsh), residual=res if with_res else None, gn_out=32)
            monkeypatch.setenv("CVVAE_CONV_FORCE", "1x8x32:2x4x1:2")
            y8, p8 = ops.conv(x, pw2, **kw)
            monkeypatch.setenv("CVVAE_CONV_FORCE", "1x8x32:1x4x1:2")
            y4, p4 = ops.conv(x, pw2, **kw)
            monkeypatch.delenv("CVVAE_CONV_FORCE")
            assert "w1x4x1" in names[-1] and "w2x4x1" in names[-2], names
            if with_res:
                assert (y8.float() - y4.float()).abs().max().item() <= (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * 8
            else:
                assert torch.equal(y8, y4)
        ones, zeros = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
        a, b = ops.gn_finalize(p8, ones, zeros, 1e-6), ops.gn_finalize(p4, ones, zeros, 1e-6)
        assert torch.allclose(a[0], b[0], rtol=1e-5) and torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-6)
        w3 = (torch.randn((128, 128, 3, 3, 3), generator=g) / (128 * 27) ** 0.5).to(dtype).cuda()
        pw3 = ops.pack_weight_tfolds(w3, torch.randn(128, generator=g).cuda())
        for pad in (((2, 0), (1, 1), (1, 1)), ((1, 1), (1, 1), (1, 1))):
            kw = dict(pad=pad, pad_mode_t=REP, pad_mode_hw=REP, prologue=1, gn=(gsc, gsh), gn_out=32)
            y8, _ = ops.conv(x, pw3, **kw)
            monkeypatch.setenv("CVVAE_CONV_FORCE", "2x8x16:1x4x1:1")
            y4, _ = ops.conv(x, pw3, **kw)
            monkeypatch.delenv("CVVAE_CONV_FORCE")
            if "t2x8x16_w1x4x1" not in names[-1]:
                break  # (the 3x3x3 four-wave forms are slower than the 8-wave tile and built only with make NW4=1)
            # a two-frame tile of the 4-wave instance whose frames have different fold plans (frames 0, 1) walks the plain
            # three time groups where the 8-wave instance multiplies the summed slots: same value up to the rounding of the
            # folded weights there, bit-identical everywhere else
            assert torch.equal(y8[:, 2:], y4[:, 2:]), f"pad {pad}"
            assert (y8[:, :2].float() - y4[:, :2].float()).abs().max().item() <= (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * 8
            # phase order of the wave groups (tuning knob): scheduling only, bit-identical results
            monkeypatch.setenv("CVVAE_CONV_PHASE_SYNC", "1")
            ys, _ = ops.conv(x, pw3, **kw)
            monkeypatch.delenv("CVVAE_CONV_PHASE_SYNC")
            assert torch.equal(y8, ys)
    finally:
        ops.PROFILE = None


@pytest.mark.parametrize("family", ["sd3", "vae3d"])
def test_fp32_model_meets_north_star_tolerance(family, golden_dir):
    """fp32 models (the reference's default: from_pretrained without torch_dtype) run in split precision -- fp32 tensors in HBM,
    every product as three fp16 MFMAs.  Against the reference's fp32 CPU outputs (tests/golden): latent max |delta| <= 1e-3
    (north_star's bound, as a MAXIMUM) with a wide margin, for clips, images, two temporal windows and blended spatial tiles."""
    from oracle import parity as P
    from oracle.golden_cases import CASES
    for name in sorted(n for n in CASES if n.startswith(family)):
        fam, over, shape, wseed, xseed = CASES[name]
        m, _ = build(fam, over, torch.float32, wseed)
        r = P.measure(m, name, golden_dir)
        print("\n" + P.fmt("f32", r))
        assert r["moments_max_abs"] <= 1e-4 and r["recon_max_abs"] <= 5e-4 and r["recon_psnr_db"] >= 90.0, r


def test_fp32_forward_and_u8_paths():
    """fp32 model through the public API: forward(), encode_frames_u8 / decode_to_frames_u8 (the scripts' arithmetic in fp32)."""
    m, sd = build("sd3", {}, torch.float32, 2)
    g = torch.Generator().manual_seed(4)
    u8 = torch.randint(0, 256, (5, 64, 96, 3), generator=g, dtype=torch.uint8)
    video = (u8.permute(3, 0, 1, 2).unsqueeze(0).float() / 127.5 - 1.0).cuda()
    z = m.encode(video).latent_dist.parameters
    assert z.dtype == torch.float32 and torch.equal(m.encode_frames_u8(u8.cuda()).latent_dist.parameters, z)
    with torch.no_grad():
        ref = O.encode_moments(video.cpu(), sd, {}, "sd3")
    assert (z.cpu() - ref).abs().max() <= 1e-4
    y = m(video).sample
    want = ((torch.clamp(y.squeeze(0).permute(1, 2, 3, 0), -1.0, 1.0) + 1.0) * 127.5).to("cpu", dtype=torch.uint8)
    assert torch.equal(m.decode_to_frames_u8(z[:, :16]).cpu(), want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_gn_silu_apply_pass_is_bit_identical_to_the_fused_prologue(dtype, monkeypatch):
    """cvvae_gn_silu_apply + conv without prologue == conv with the fused GN+SiLU prologue (same fp32 arithmetic, one rounding):
    bit for bit at op level for the 16-bit dtypes; through a whole model (CVVAE_PREPASS=1 vs 0) up to the summation order of the
    differently tiled instances.  (Tuning aid: the fused form is the product path -- the pass measured 3-8 % SLOWER end to end.)"""
    from cvvae_amd import ops
    g = torch.Generator().manual_seed(21)
    x = (torch.randn((2, 3, 24, 40, 128), generator=g) * 1.5 + 0.2).to(dtype).cuda()
    gsc = (1.0 + 0.1 * torch.randn((2, 128), generator=g)).cuda()
    gsh = (0.1 * torch.randn((2, 128), generator=g)).cuda()
    w = (torch.randn((256, 128, 3, 3, 3), generator=g) / (128 * 27) ** 0.5).to(dtype).cuda()
    pw = ops.pack_weight_tfolds(w, torch.randn(256, generator=g).cuda())
    kw = dict(pad=((2, 0), (1, 1), (1, 1)), pad_mode_t=REP, pad_mode_hw=REP)
    # (the same tile for both: since round 6 the prologue form prefers 32-channel K chunks, another summation order)
    monkeypatch.setenv("CVVAE_CONV_FORCE", "1x8x32:1x8x1:1")
    fused = ops.conv(x, pw, prologue=1, gn=(gsc, gsh), **kw)
    xa = ops.gn_silu_apply(x, (gsc, gsh))
    unfused = ops.conv(xa, pw, **kw)
    monkeypatch.delenv("CVVAE_CONV_FORCE")
    if dtype == torch.float32:  # (two compilations of the same fp32 expression: last-bit differences of exp / rcp scheduling)
        assert (unfused - fused).abs().max().item() <= 2e-6 * fused.abs().max().item()
    else:
        assert torch.equal(unfused, fused)
    ref = torch.nn.functional.silu(x.float() * gsc.view(2, 1, 1, 1, 128) + gsh.view(2, 1, 1, 1, 128))
    assert (xa.float() - ref).abs().max().item() <= (4e-6 if dtype == torch.float32 else (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11)) * ref.abs().max().item()
    m, _ = build("sd3", {}, dtype, 3)
    xin = seeded_input((1, 3, 5, 64, 64), 8).to(dtype).cuda()
    monkeypatch.setenv("CVVAE_PREPASS", "0")
    z0 = m.encode(xin).latent_dist.parameters
    y0 = m.decode(z0[:, :16]).sample
    monkeypatch.setenv("CVVAE_PREPASS", "1")
    z1 = m.encode(xin).latent_dist.parameters
    y1 = m.decode(z0[:, :16]).sample
    # through a whole model the conv instances differ (no-prologue instances are selected: other tile shapes / K-group splits,
    # i.e. another summation order), so the comparison is at the storage dtype's noise level, not bit for bit
    tolz, toly = {torch.float32: (2e-5, 1e-4), torch.float16: (4e-3, 1.5e-2), torch.bfloat16: (3.5e-2, 1e-1)}[dtype]
    assert (z0.float() - z1.float()).abs().max().item() <= tolz and (y0.float() - y1.float()).abs().max().item() <= toly


def test_packed_weight_cache(tmp_path, monkeypatch):
    """save_packed_weights / load_packed_weights: a second model instance of the same checkpoint installs the
    packed MFMA-order weights from the file and produces the same bits without launching a single pack kernel"""
    from cvvae_amd import ops
    dtype = torch.float16
    m, _ = build("sd3", {}, dtype, 4)
    x = seeded_input((1, 3, 5, 64, 64), 5).to(dtype).cuda()
    z = m.encode(x).latent_dist.mode()
    y = m.decode(z).sample
    path = str(tmp_path / "packed.pt")
    n = m.save_packed_weights(path)
    assert n > 60
    m2, _ = build("sd3", {}, dtype, 4)
    assert m2.load_packed_weights(path) == n

    def no_pack(*a, **k):
        raise AssertionError("a weight was packed although the cache holds it")
    for name in ("pack_weight", "pack_weight_tfolds", "pack_weight_t1", "pack_weight_upfold"):
        monkeypatch.setattr(ops, name, no_pack)
    assert torch.equal(m2.encode(x).latent_dist.mode(), z) and torch.equal(m2.decode(z).sample, y)
    monkeypatch.undo()
    m3, _ = build("sd3", {}, torch.bfloat16, 4)  # another dtype: nothing matches, everything is packed as usual
    assert m3.load_packed_weights(path) == 0
