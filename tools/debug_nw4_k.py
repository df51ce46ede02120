"""debug aid: expose the statistics shift K of every record (CVVAE_STATS_NOSHIFT=2) and compare it with the stored value it is
defined as (first pixel of the tile, first channel of the slot)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVVAE_STATS_NOSHIFT"] = "2"
os.environ["CVVAE_CONV_FORCE"] = "1x8x32:1x4x1:2"
import torch
from cvvae_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16
x = torch.randn((3, 5, 64, 96, 128), device="cuda").to(dt)
gsc = (1 + 0.1 * torch.randn((3, 128), device="cuda")); gsh = 0.1 * torch.randn((3, 128), device="cuda")
w = (torch.randn((128, 128, 1, 3, 3), device="cuda") / (128 * 9) ** 0.5).to(dt)
pw = ops.pack_weight(w.reshape(128, 128, 9), torch.randn(128, device="cuda"), (1, 3, 3))
for it in range(6):
    y, part = ops.conv(x, pw, pad=((0, 0), (1, 1), (1, 1)), prologue=1, gn=(gsc, gsh), gn_out=32)
    torch.cuda.synchronize()
    K = part.buf[..., 2]                                  # [B, tiles, 32 groups]
    B, T, H, W, C = y.shape
    y0 = y.float().view(B, T, H // 8, 8, W // 32, 32, C)[:, :, :, 0, :, 0, :]      # first pixel of every tile: [B,T,8,3,C]
    exp = y0.reshape(B, T * (H // 8) * (W // 32), 32, 4)[..., 0]                   # first channel of every 4-channel slot
    bad = (K != exp).nonzero()
    print(f"run {it}: records {K.numel()}, K != expected: {bad.shape[0]}")
    for b, s, g in bad[:8].tolist():
        t_, r = divmod(s, 24); th, tw = divmod(r, 3)
        print(f"   row {b} tile {s} (t {t_} th {th} tw {tw}) group {g}: K {K[b, s, g].item():.6f} expected {exp[b, s, g].item():.6f}")
