import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVVAE_STATS_NOSHIFT"] = "3"
os.environ["CVVAE_CONV_FORCE"] = "1x8x32:1x4x1:2"
import torch
from cvvae_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16
x = torch.randn((3, 5, 64, 96, 128), device="cuda").to(dt)
gsc = (1 + 0.1 * torch.randn((3, 128), device="cuda")); gsh = 0.1 * torch.randn((3, 128), device="cuda")
w = (torch.randn((128, 128, 1, 3, 3), device="cuda") / (128 * 9) ** 0.5).to(dt)
pw = ops.pack_weight(w.reshape(128, 128, 9), torch.randn(128, device="cuda"), (1, 3, 3))
tot = 0
for it in range(40):
    y, part = ops.conv(x, pw, pad=((0, 0), (1, 1), (1, 1)), prologue=1, gn=(gsc, gsh), gn_out=32)
    torch.cuda.synchronize()
    D = part.buf[..., 2]
    bad = (D != 0).nonzero()
    tot += bad.shape[0]
    for b, s, g in bad[:4].tolist():
        print(f"run {it}: row {b} tile {s} group {g}: sum |K_lane - K_writer| = {D[b, s, g].item():.6f}  (mean field {part.buf[b, s, g, 1].item():.5f})")
print("records with inconsistent shifts over 40 runs:", tot)
