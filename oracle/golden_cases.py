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
