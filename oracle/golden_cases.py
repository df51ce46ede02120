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

