#!/usr/bin/env python
"""Debug aid: run golden cases with a device sync + print after every C-ABI conv launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import cvvae_amd
from cvvae_amd import ops
from oracle.golden_cases import CASES
from oracle.seeded import seeded_input, seeded_state_dict

def obs(d, pw, launch):
    print("  conv", ops.conv_kernel_name(d), "B,T,H,W,Cin", d.B, d.Ti, d.Hi, d.Wi, d.Cin, "->", d.To, d.Ho, d.Wo, d.Cout,
          "k", d.kT, d.kH, d.kW, "s", d.sT, d.sH, d.sW, "pro", d.prologue, "rpb", d.gn_rows_per_batch, "om", d.out_mode, flush=True)
    launch()
    torch.cuda.synchronize()

ops.PROFILE = obs
names = sys.argv[1:] or sorted(CASES)
for dtype in (torch.float16, torch.bfloat16):
    for name in names:
        family, over, shape, wseed, xseed = CASES[name]
        print("CASE", name, dtype, flush=True)
        cls = cvvae_amd.CVVAESD3Model if family == "sd3" else cvvae_amd.CVVAEModel
        m = cls(**over)
        sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, wseed)
        m.load_state_dict(sd, strict=True)
        m = m.to(dtype).cuda().eval()
        x = seeded_input(shape, xseed).to(dtype).cuda()
        mom = m.encode(x).latent_dist.parameters
        torch.cuda.synchronize()
        print(" encode ok", flush=True)
        gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        zc = gold["moments"].shape[1] // 2
        z = torch.from_numpy(gold["moments"][:, :zc]).to(dtype).cuda()
        rec = m.decode(z).sample
        torch.cuda.synchronize()
        print(" decode ok", flush=True)
