"""Golden-vector case table shared by oracle/make_golden.py (generator) and the tests.  TEST INFRASTRUCTURE."""

# name -> (family, config overrides, input shape [B,3,T,H,W], weight seed, input seed)
CASES = {
    "sd3_t5_64": ("sd3", {}, (1, 3, 5, 64, 64), 0, 0),
    "sd3_t1_64": ("sd3", {}, (1, 3, 1, 64, 64), 0, 1),            # image mode (Appendix A.15)
    "sd3_t33_32": ("sd3", {}, (1, 3, 33, 32, 32), 0, 2),          # two temporal windows
    "sd3_tiled_t5_160x200": ("sd3", {"tile_spatial_size": 144}, (1, 3, 5, 160, 200), 0, 3),  # 2x2 spatial tiles + blend
    "vae3d_t5_64": ("vae3d", {}, (1, 3, 5, 64, 64), 0, 0),
    "vae3d_t1_64": ("vae3d", {}, (1, 3, 1, 64, 64), 0, 1),
    "vae3d_t33_32": ("vae3d", {}, (1, 3, 33, 32, 32), 0, 2),
    "vae3d_tiled_t5_160x200": ("vae3d", {"tile_spatial_size": 144}, (1, 3, 5, 160, 200), 0, 3),
}

# the frozen 2-D constraint decoder: name -> (config = the yaml's params, latent shape, weight seed, latent seed)
CONSTRAINT_CFG = dict(in_channels=16, out_channels=3, up_block_types=["UpDecoderBlock2D"] * 4,
                      block_out_channels=[128, 256, 512, 512], layers_per_block=2, norm_num_groups=32, act_fn="silu",
                      mid_block_add_attention=True)  # configs/cvvae_sd3_constraint_training.yaml:40-51
CONSTRAINT_CASES = {
    "constraint2d_t3_8": (CONSTRAINT_CFG, (1, 16, 3, 8, 8), 0, 10),     # 5-D latents: decoded frame by frame
    "constraint2d_4d_12x8": (CONSTRAINT_CFG, (2, 16, 12, 8), 0, 11),    # 4-D latents, two images, non-square
}


# the frozen 2-D halves of the SD2.1-compatible family (lvdm/modules/diffusionmodules/model.py:775-887):
# name -> (class name, constructor kwargs, input shape, weight seed, input seed)
LDM_CFG = dict(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3,
               resolution=256, z_channels=4, double_z=True)   # the SD2.1 image VAE (first_stage_config of the SD2.1 family)
LDM_CASES = {
    "ldm2d_enc_t3_32": ("EncoderWith3DWrapper", LDM_CFG, (1, 3, 3, 32, 32), 0, 20),    # clip: every frame on its own
    "ldm2d_enc_4d_24x16": ("EncoderWith3DWrapper", LDM_CFG, (2, 3, 24, 16), 0, 21),    # two images, non-square
    "ldm2d_dec_t3_8": ("DecoderWith3DWrapper", LDM_CFG, (1, 4, 3, 8, 8), 0, 22),
    "ldm2d_dec_4d_6x4": ("DecoderWith3DWrapper", LDM_CFG, (2, 4, 6, 4), 0, 23),
}


# BASELINE.json configs at FULL size (SURVEY 8c "and on GPU the BASELINE shapes"): fixtures hold the full `moments` and a
# bounded part of `recon` -- recon_sub[t] = recon[:, :, t, (t % s)::s, (3t % s)::s] (stride s over H and W with a per-frame phase,
# so every kernel tile and every pixel phase of the frame is sampled) -- plus fp64 global moments of the whole recon.
# name -> (family, config overrides, input shape, weight seed, input seed, recon stride s)
BIG_CASES = {
    "cfg1_vae3d_t1_256": ("vae3d", {}, (1, 3, 1, 256, 256), 0, 21, 1),      # BASELINE cfg 1: one 256x256 image
    "cfg2_vae3d_t17_256": ("vae3d", {}, (1, 3, 17, 256, 256), 0, 22, 4),    # BASELINE cfg 2
    "cfg3_sd3_t17_512": ("sd3", {}, (1, 3, 17, 512, 512), 0, 23, 4),        # BASELINE cfg 3: the configuration the metric is quoted on
    # one 17-frame window of BASELINE cfg 4 (720x1280: 2x3 spatial tiles of 576/272 x 576/576/384 pixels, blended)
    "cfg4win_sd3_t17_720x1280": ("sd3", {}, (1, 3, 17, 720, 1280), 0, 24, 8),
}

# ENCODE-ONLY fixtures at full size (the training-side latent pre-compute of BASELINE cfg 5: batch-8 T=33 512x512 encode; the
# fixture is a B = 2 slice of the 8 -- batch items are independent network calls, two of them pin the batch indexing -- with both
# 17-frame windows of every clip): the posterior mean in full, the log-variance at stride 2 over H and W
# name -> (family, config overrides, input shape, weight seed, input seed[, stride s of the stored posterior mean over H and W --
# latent frame t sampled at rows (t % s)::s, columns (3t % s)::s as recon_subsample does; default 1 = in full])
ENC_CASES = {
    "cfg5slice_sd3_b2_t33_512_enc": ("sd3", {}, (2, 3, 33, 512, 512), 0, 25),
    # BASELINE cfg 4 WHOLE: T = 129 at 720x1280 = 8 temporal windows x 6 blended spatial tiles (48 encoder calls): the full
    # window / tile / blend wrapper at the size it was written for
    "cfg4_sd3_t129_720x1280_enc": ("sd3", {}, (1, 3, 129, 720, 1280), 0, 26, 2),
}

# DECODE-ONLY fixtures at full size: a SEEDED latent (the decoder is judged on a given input; it need not be an encoder's output)
# through the whole decode wrapper; the reconstruction is stored at stride s over H and W with recon_subsample's per-frame phase.
# name -> (family, config overrides, latent shape, weight seed, latent seed, recon stride s)
DEC_CASES = {
    # BASELINE cfg 4's decode side WHOLE: 33 latent frames at 90x160 -> 129 frames at 720x1280: 8 temporal windows x 6 blended tiles
    "cfg4_sd3_z33_90x160_dec": ("sd3", {}, (1, 16, 33, 90, 160), 0, 27, 16),
}


# BACKWARD fixtures at the size tools/train_step_bench.py times (round 5): the reference's OWN Encoder3D / Decoder3D (the full
# 4-level, layers_per_block = 2 networks of CVVAESD3Model) under torch.autograd on one 17-frame 256x256 crop, loss = <output, seeded
# cotangent>.  Stored: the forward output and dL/d(input) at stride s (recon_subsample's per-frame phase), and per parameter tensor
# the gradient's L2 norm plus a seeded sample of <= 4096 elements (grad_sample_index).
# name -> (family, config overrides, pixel shape, weight seed, input seed, cotangent seeds (enc, dec), latent seed, strides (dx, dz))
GRAD_CASES = {
    "grad_sd3_t17_256": ("sd3", {}, (1, 3, 17, 256, 256), 0, 31, (32, 33), 34, (4, 1)),
    # the other family (models/modeling_vae.py's vae3d networks: temporal attention in the mid blocks, 4 latent channels)
    "grad_vae3d_t17_256": ("vae3d", {}, (1, 3, 17, 256, 256), 0, 41, (42, 43), 44, (4, 1)),
}


def grad_sample_index(name: str, numel: int, n: int = 4096):
    """the flat indices of parameter `name`'s stored gradient sample: all of them for tensors of <= n elements, else n seeded ones"""
    import zlib

    import torch
    if numel <= n:
        return torch.arange(numel)
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(name.encode()) + 977)
    return torch.randint(0, numel, (n,), generator=g)


def recon_subsample(recon, s: int):
    """recon [B,C,T,H,W] (numpy or torch) -> [B,C,T,H//s,W//s]: frame t sampled at rows (t % s)::s, columns (3t % s)::s (cropped to
    the common H//s x W//s when s does not divide the frame: the phases would otherwise differ in length)"""
    import numpy as np
    hs, ws = recon.shape[3] // s, recon.shape[4] // s
    frames = [recon[:, :, t, (t % s)::s, ((3 * t) % s)::s][..., :hs, :ws] for t in range(recon.shape[2])]
    if hasattr(recon, "numpy"):
        import torch
        return torch.stack(frames, dim=2)
    return np.stack(frames, axis=2)
