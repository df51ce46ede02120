"""CPU check of the HOST logic above the C ABI -- the window / tile wrappers of modeling.py and the launch programs of engine.py --
against the reference's golden vectors: the real CVVAEModel / CVVAESD3Model classes run end to end with every kernel replaced by a
plain-PyTorch emulation of its documented arithmetic (tests/emu_ops.py).  What this pins without a GPU: which tensor, weight form
(plain / time folds / single-frame fold / folded upsample / fused shortcut), padding, prologue, epilogue and output mode every launch
gets; the 17-frame windows, the 576/448 tiles and their blends; the posterior; the u8 pre/post-processing plumbing.  The kernels
themselves are compared with the same fixtures on the GPU (tests/test_gpu_model.py)."""
import os

import numpy as np
import pytest
import torch

from oracle.golden_cases import CASES, CONSTRAINT_CASES
from oracle.seeded import seeded_input, seeded_state_dict
from tests import emu_ops

TOL = 2e-4  # fp32 against the reference's fp32 fixtures: summation order only (measured <= 6e-5)


_SD = {}


def build(family, over, wseed):
    import cvvae_amd
    cls = cvvae_amd.CVVAESD3Model if family == "sd3" else cvvae_amd.CVVAEModel
    m = cls(**over)
    if (family, wseed) not in _SD:  # the seeded weights depend on names and shapes only
        _SD[(family, wseed)] = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, wseed)
    m.load_state_dict(_SD[(family, wseed)], strict=True)
    return m.eval()


@pytest.mark.parametrize("name", sorted(n for n in CASES if n != "vae3d_tiled_t5_160x200"))  # (tiles: the sd3 case; 18 s apiece)
def test_models_match_reference_golden_through_emulated_kernels(name, golden_dir):
    family, over, shape, wseed, xseed = CASES[name]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    m = build(family, over, wseed)
    x = seeded_input(shape, xseed)
    with emu_ops.patched(whole_model=True), torch.no_grad():
        post = m.encode(x).latent_dist
        rec = m.decode(post.mode()).sample
    assert np.abs(post.parameters.numpy() - gold["moments"]).max() <= TOL
    assert np.abs(rec.numpy() - gold["recon"]).max() <= TOL


@pytest.mark.parametrize("name,env", [("sd3_t1_64", {"CVVAE_FOLD_T1": "0"}), ("sd3_t1_64", {"CVVAE_FOLD_UPSAMPLE": "0"}),
                                      ("vae3d_t5_64", {"CVVAE_FUSE_SHORTCUT": "0"}), ("sd3_t5_64", {"CVVAE_FOLD_TIME": "0"}),
                                      ("vae3d_t5_64", {"CVVAE_PREPASS": "1"}), ("sd3_t5_64", {"CVVAE_FOLD_UPSAMPLE": "0"})])
def test_alternative_launch_programs_agree(name, env, golden_dir, monkeypatch):
    """every tuning switch of engine.py selects another launch sequence for the same arithmetic"""
    family, over, shape, wseed, xseed = CASES[name]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m = build(family, over, wseed)
    with emu_ops.patched(whole_model=True), torch.no_grad():
        post = m.encode(seeded_input(shape, xseed)).latent_dist
        rec = m.decode(post.mode()).sample
    assert np.abs(post.parameters.numpy() - gold["moments"]).max() <= TOL and np.abs(rec.numpy() - gold["recon"]).max() <= TOL


@pytest.mark.parametrize("name", sorted(CONSTRAINT_CASES))
def test_constraint_decoder_matches_reference_golden_through_emulated_kernels(name, golden_dir):
    from cvvae_amd.constraint import DecoderWith3DWrapper
    cfg, zshape, wseed, zseed = CONSTRAINT_CASES[name]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    m = DecoderWith3DWrapper(**cfg)
    m.load_state_dict(seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, wseed), strict=True)
    m = m.eval().requires_grad_(False)
    with emu_ops.patched(whole_model=True), torch.no_grad():
        rec = m(seeded_input(zshape, zseed))
    assert np.abs(rec.numpy() - gold["recon"]).max() <= TOL


def test_forward_latents_and_u8_plumbing():
    """forward(), encode_latents() and the u8 entry points are compositions of encode / decode: same numbers"""
    m = build("sd3", {}, 4)
    x = seeded_input((2, 3, 5, 32, 32), 12)
    with emu_ops.patched(whole_model=True), torch.no_grad():
        post = m.encode(x).latent_dist
        assert torch.equal(m(x).sample, m.decode(post.mode()).sample)
        g = torch.Generator().manual_seed(3)
        zs = post.sample(generator=torch.Generator().manual_seed(3))
        assert torch.equal(m(x, sample_posterior=True, generator=g, return_dict=False)[0], m.decode(zs).sample)
        lat = m.encode_latents(x, sample=False, n_samples_a_time=1, scale_factor=0.5)
        assert torch.allclose(lat, 0.5 * post.mode(), atol=1e-4)  # (per-sample rounds: another summation order on the CPU)
        # u8 frames: the scripts' normalisation, then encode; decode, then the scripts' clamp/scale/u8
        frames = (torch.rand(5, 32, 32, 3, generator=torch.Generator().manual_seed(1)) * 255).to(torch.uint8)
        xs = (frames.permute(3, 0, 1, 2).unsqueeze(0).float() / 127.5 - 1.0)
        zu = m.encode_frames_u8(frames).latent_dist.mode()
        assert torch.allclose(zu, m.encode(xs).latent_dist.mode(), atol=1e-5)
        out = m.decode_to_frames_u8(zu)
        ref = ((torch.clamp(m.decode(zu).sample[0], -1, 1) + 1) * 127.5).to(torch.uint8).permute(1, 2, 3, 0)
        assert out.dtype == torch.uint8 and tuple(out.shape) == (5, 32, 32, 3) and torch.equal(out, ref)


def test_packed_weight_cache_roundtrip(tmp_path, monkeypatch):
    """save_packed_weights / load_packed_weights: a second instance of the same checkpoint installs the packed forms
    and runs without packing anything; a changed parameter falls back to packing for its layer only"""
    from cvvae_amd import ops
    m = build("sd3", {}, 4)
    x = seeded_input((1, 3, 5, 32, 32), 2)
    path = str(tmp_path / "packed.pt")
    with emu_ops.patched(whole_model=True), torch.no_grad():
        z = m.encode(x).latent_dist.mode()
        y = m.decode(z).sample
        n = m.save_packed_weights(path)
        assert n > 60
        m2 = build("sd3", {}, 4)
        assert m2.load_packed_weights(path) == n

        def no_pack(*a, **k):
            raise AssertionError("a weight was packed although the cache holds it")
        for name in ("pack_weight", "pack_weight_tfolds", "pack_weight_t1", "pack_weight_upfold"):
            monkeypatch.setattr(ops, name, no_pack)
        assert torch.equal(m2.encode(x).latent_dist.mode(), z) and torch.equal(m2.decode(z).sample, y)
        monkeypatch.undo()
    with emu_ops.patched(whole_model=True), torch.no_grad():
        m3 = build("sd3", {}, 4)
        m3.encoder.conv_in.weight.data.mul_(1.5)
        assert m3.load_packed_weights(path) == n - 1
        assert not torch.equal(m3.encode(x).latent_dist.mode(), z)
        import cvvae_amd
        with pytest.warns(UserWarning):  # another model class: ignored before anything is read
            assert cvvae_amd.CVVAEModel().load_packed_weights(path) == 0


def test_packed_weight_cache_skips_stale_entries(tmp_path):
    """a warm-up pass, THEN new weights (load_state_dict / in-place edit), THEN save_packed_weights must not store the
    old packed forms under the new parameters' fingerprints: stale cache entries are left out of the export, and a fresh model
    that imports the file computes with the NEW weights."""
    m = build("sd3", {}, 4)
    x = seeded_input((1, 3, 5, 32, 32), 2)
    path = str(tmp_path / "packed.pt")
    with emu_ops.patched(whole_model=True), torch.no_grad():
        z_old = m.encode(x).latent_dist.mode()                   # packs everything the encoder uses
        n_all = len(m.encoder._cache().export_packed())
        m.encoder.conv_in.weight.mul_(1.5)                       # in-place edit after the warm-up (what load_state_dict does)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        exp = m.encoder._cache().export_packed()
        assert len(exp) == n_all - 1 and not any(t.startswith("conv_in") for t in exp)
        m.save_packed_weights(path)
        z_new = m.encode(x).latent_dist.mode()
        assert not torch.equal(z_new, z_old)
        m2 = build("sd3", {}, 4)
        m2.load_state_dict(sd)
        m2.load_packed_weights(path)
        assert torch.equal(m2.encode(x).latent_dist.mode(), z_new)


def test_encode_latents_flatten_frames_matches_reference_layout():
    """DiffusionEngineFor3DVAE.encode_first_stage returns '(b t) c h w' for clips as well (diffusion.py:380-385)"""
    m = build("sd3", {}, 4)
    x = seeded_input((2, 3, 5, 32, 32), 3)
    with emu_ops.patched(whole_model=True), torch.no_grad():
        z5 = m.encode_latents(x, sample=False)
        z4 = m.encode_latents(x, sample=False, flatten_frames=True)
    assert z5.dim() == 5 and z4.dim() == 4 and z4.shape[0] == z5.shape[0] * z5.shape[2]
    assert torch.equal(z4, z5.permute(0, 2, 1, 3, 4).reshape(z4.shape))


def test_fp32_fused_shortcut_falls_back_when_scale_would_overflow():
    """the fused 1x1 shortcut of an fp32 model shares conv2's power-of-two pack scale; with shortcut weights 64x
    larger than conv2's the fp16 hi part would overflow -- the block must then run unfused, with the same result"""
    from cvvae_amd import engine
    m = build("sd3", {}, 4).float()
    x = seeded_input((1, 3, 5, 32, 32), 2)
    blk = m.encoder.down_blocks[1].resnets[0]
    with torch.no_grad():
        blk.conv_shortcut.weight.mul_(4096.0 / float(blk.conv_shortcut.weight.abs().max()) * float(blk.conv2.weight.abs().max()))
    wc = m.encoder._cache()
    calls = []
    with emu_ops.patched(whole_model=True), torch.no_grad():
        real = emu_ops.conv

        def spy(xx, pw, **kw):
            calls.append(kw.get("shortcut") is not None)
            return real(xx, pw, **kw)
        import cvvae_amd.ops as ops_mod
        old = ops_mod.conv
        ops_mod.conv = spy
        try:
            m.encode(x)
        finally:
            ops_mod.conv = old
        pw2 = wc.conv("down_blocks.1.resnets.0.conv2", (1, 3, 3))
        assert not engine._shortcut_scale_fits(wc, "down_blocks.1.resnets.0.conv_shortcut", pw2, torch.float32)
        assert engine._shortcut_scale_fits(wc, "down_blocks.2.resnets.0.conv_shortcut",
                                           wc.conv("down_blocks.2.resnets.0.conv2", (1, 3, 3)), torch.float32)
    assert sum(calls) == 1, "only the other channel-changing block keeps its fused shortcut"


def _ldm_model(name):
    import cvvae_amd.constraint_ldm as C
    from oracle.golden_cases import LDM_CASES
    cls, cfg, shape, wseed, xseed = LDM_CASES[name]
    m = getattr(C, cls)(**cfg)
    m.load_state_dict(seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, wseed), strict=True)
    return m.eval().requires_grad_(False), seeded_input(shape, xseed)


@pytest.mark.parametrize("name", ["ldm2d_enc_t3_32", "ldm2d_enc_4d_24x16", "ldm2d_dec_t3_8", "ldm2d_dec_4d_6x4"])
def test_ldm_2d_wrappers_match_reference_golden_through_emulated_kernels(name, golden_dir):
    """EncoderWith3DWrapper / DecoderWith3DWrapper of the SD2.1-compatible family (model.py:775-887) -- the launch
    programs (per-frame rows, zero-pad downsample, folded upsample, quant / post_quant 1x1 layers, middle attention) against fixtures
    produced by the reference's own classes (oracle/make_golden.py ldm)"""
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    m, x = _ldm_model(name)
    with emu_ops.patched(whole_model=True), torch.no_grad():
        out = m(x)
    assert out.shape == gold["out"].shape
    assert np.abs(out.numpy() - gold["out"]).max() <= TOL


def test_ldm_2d_wrappers_import_path_and_state_dict():
    """the reference's module path and, when it is mounted, its classes' state-dict names and shapes"""
    from lvdm.modules.diffusionmodules.model import DecoderWith3DWrapper, EncoderWith3DWrapper
    from oracle.golden_cases import LDM_CFG
    from oracle.ref_loader import load_reference_ldm, reference_available
    e, d = EncoderWith3DWrapper(**LDM_CFG), DecoderWith3DWrapper(**LDM_CFG)
    assert len(e.state_dict()) == 108 and len(d.state_dict()) == 140
    assert "quant_conv.weight" in e.state_dict() and "post_quant_conv.weight" in d.state_dict()
    assert "quant_conv.weight" not in EncoderWith3DWrapper(legacy=False, **LDM_CFG).state_dict()
    if reference_available():
        ref = load_reference_ldm()
        for mine, theirs in ((e, ref.EncoderWith3DWrapper(**LDM_CFG)), (d, ref.DecoderWith3DWrapper(**LDM_CFG))):
            assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == {k: tuple(v.shape) for k, v in theirs.state_dict().items()}
    with pytest.raises(NotImplementedError):
        EncoderWith3DWrapper(**dict(LDM_CFG, attn_resolutions=[32]))


@pytest.mark.parametrize("name", ["sd3_t5_64", "vae3d_t5_64"])
def test_fast_fp32_fp6_launch_plumbing(name, golden_dir, monkeypatch):
    """fp32_mode = "fast": the convs behind a GroupNorm + SiLU prologue take the fp6-correction form (CVVAE_F32Q6) with a bound
    derived from the norm's affine -- every such launch carries a bound, the seeded operands stay inside it, everything else stays
    on the bf8 form (CVVAE_F32Q) or the three-MFMA one; CVVAE_F32_FP6=0 switches the fp6 form off"""
    from cvvae_amd import _lib as L
    family, over, shape, wseed, xseed = CASES[name]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    for fp6 in ("1", "0"):
        monkeypatch.setenv("CVVAE_F32_FP6", fp6)
        m = build(family, over, wseed)
        m.fp32_mode = "fast"
        del emu_ops.Q6_CALLS[:]
        with emu_ops.patched(whole_model=True), torch.no_grad():
            post = m.encode(seeded_input(shape, xseed)).latent_dist
            rec = m.decode(post.mode()).sample
        assert np.abs(post.parameters.numpy() - gold["moments"]).max() <= TOL and np.abs(rec.numpy() - gold["recon"]).max() <= TOL
        dts = {}
        for sub in (m.encoder, m.decoder):
            for tag, ent in sub._wc._c.items() if hasattr(sub, "_wc") else []:
                if len(ent) >= 3 and hasattr(ent[1], "dt"):
                    dts.setdefault(ent[1].dt, []).append(tag)
        if fp6 == "1":
            assert emu_ops.Q6_CALLS and all(0.0 < mx <= bound for mx, bound in emu_ops.Q6_CALLS), emu_ops.Q6_CALLS[:4]
            if dts:
                assert dts.get(L.F32Q6) and all(t.endswith("#q6") for t in dts[L.F32Q6])
                assert all(".conv1" in t or ".conv2" in t for t in dts[L.F32Q6])  # resnet convs only
        else:
            assert not emu_ops.Q6_CALLS and L.F32Q6 not in dts


def test_fp6_form_is_refused_when_the_norm_affine_spreads_over_many_binades():
    """one scale per launch cannot serve channels whose bounds differ by more than 8x: such a norm's convs keep the bf8 form"""
    from cvvae_amd import _lib as L
    family, over, shape, wseed, xseed = CASES["sd3_t5_64"]
    m = build(family, over, wseed)
    m.fp32_mode = "fast"
    with emu_ops.patched(whole_model=True), torch.no_grad():
        z = m.encode(seeded_input(shape, xseed)).latent_dist.mode()
        wc = m.decoder._cache()
        name = "mid_block.resnets.0"
        assert wc.act_bound(name + ".norm1") > 0.0
        assert wc.conv(name + ".conv1", (3, 3, 3), time_folds=True, act_norm=name + ".norm1").dt == L.F32Q6
        with torch.no_grad():
            m.decoder.get_parameter(name + ".norm1.weight")[3] *= 100.0  # one loud channel
        assert wc.act_bound(name + ".norm1") == 0.0
        assert wc.conv(name + ".conv1", (3, 3, 3), time_folds=True, act_norm=name + ".norm1").dt == L.F32Q
        m.decode(z)  # and the whole decoder still runs (mixed forms)


def test_ema_swap_after_a_plain_pass_in_the_same_mode_is_seen():
    """The reference's validation_step / log_images run a plain pass and THEN enter ema_scope() without another train() / eval() call
    (lvdm/models/autoencoder.py:379-384, 1193, 1426); LitEma.copy_to / restore write every parameter with requires_grad through `.data`
    (lvdm/modules/ema.py:61-86), which does not move `_version`.  The inference branch of a network that still has such parameters
    therefore re-checks the parameter checksum on EVERY pass; a frozen network (cvvae_inference_video.py:12) is not checked per pass
    and needs refresh_weights() after a `.data` write."""
    over = {}
    m = build("sd3", over, 7)
    x = seeded_input((1, 3, 5, 32, 32), 3)
    params = [p for p in m.encoder.parameters() if p.requires_grad]
    assert params, "a freshly built model is trainable: the EMA can write it"
    with emu_ops.patched(whole_model=True), torch.no_grad():
        y_live = m.encoder(x)                      # validation_step: the plain pass (consumes the eval() transition's check)
        stored = [p.detach().clone() for p in params]
        for p in params:                           # ema_scope(): LitEma.copy_to
            p.data.mul_(0.5)
        v = [p._version for p in params]
        y_ema = m.encoder(x)                       # the EMA pass, same mode, no refresh_weights() call
        assert [p._version for p in params] == v
        assert not torch.equal(y_ema, y_live), "the EMA pass reused the live weights' packed forms"
        for p, s in zip(params, stored):           # LitEma.restore
            p.data.copy_(s)
        assert torch.equal(m.encoder(x), y_live)
        # the same through a FRESH model that holds the halved weights from the start: what the EMA pass must equal
        m2 = build("sd3", over, 7)
        for p in m2.encoder.parameters():
            p.data.mul_(0.5)
        m2.encoder.refresh_weights()
        assert torch.equal(m2.encoder(x), y_ema)
        # frozen network: no per-pass check (the staleness itself is shown on the device, tests/test_gpu_round5.py: the emulated
        # packed forms alias the parameters); the checksum still sees the write when asked
        m.requires_grad_(False)
        assert not m.encoder._ema_writable()
        m.encoder(x)
        m.encoder.conv_in.weight.data.mul_(2.0)
        assert m.encoder.refresh_weights(only_if_changed=True)
