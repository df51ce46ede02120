import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVVAE_CONV_FORCE"] = sys.argv[1] if len(sys.argv) > 1 else "1x8x32:1x4x1:2"
import torch
from cvvae_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16
x = torch.randn((3, 5, 64, 96, 128), device="cuda").to(dt)
gsc = (1 + 0.1 * torch.randn((3, 128), device="cuda")); gsh = 0.1 * torch.randn((3, 128), device="cuda")
w = (torch.randn((128, 128, 1, 3, 3), device="cuda") / (128 * 9) ** 0.5).to(dt)
pw = ops.pack_weight(w.reshape(128, 128, 9), torch.randn(128, device="cuda"), (1, 3, 3))
def stats(y):
    B, T, H, W, C = y.shape
    yt = y.float().view(B, T, H // 8, 8, W // 32, 32, 32, 4).permute(0, 1, 2, 4, 6, 3, 5, 7).reshape(B, T * (H // 8) * (W // 32), 32, -1)
    return yt.mean(-1)
tot, worst = 0, 0.0
for it in range(60):
    y, part = ops.conv(x, pw, pad=((0, 0), (1, 1), (1, 1)), prologue=1, gn=(gsc, gsh), gn_out=32)
    torch.cuda.synchronize()
    err = (part.buf[..., 1] - stats(y)).abs()
    tot += int((err > 1e-4).sum()); worst = max(worst, err.max().item())
print(f"NOSHIFT={os.environ.get('CVVAE_STATS_NOSHIFT')} force={os.environ['CVVAE_CONV_FORCE']}: wrong record means over 60 runs: {tot}, worst error {worst:.3e}")
