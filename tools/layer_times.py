#!/usr/bin/env python
"""Per-launch timing of one encode+decode step (HIP events around every conv launch): where the step's time goes, layer by
layer.  usage (GPU box): python tools/layer_times.py [--family sd3|vae3d] [--shape 1,3,17,512,512]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cvvae_amd
from cvvae_amd import ops

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--family", default="sd3", choices=["sd3", "vae3d"])
ap.add_argument("--shape", default="1,3,17,512,512")
a = ap.parse_args()
torch.manual_seed(0)
dtype = torch.bfloat16
vae = (cvvae_amd.CVVAESD3Model if a.family == "sd3" else cvvae_amd.CVVAEModel)().to(dtype).cuda().eval().requires_grad_(False)
x = (torch.rand(tuple(int(v) for v in a.shape.split(","))) * 2 - 1).to(dtype).cuda()
rec = []


def obs(d, pw, launch):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record()
    fl = 2.0 * d.B * d.To * d.Ho * d.Wo * d.Cout * (pw.cin_real * (pw.alg_taps or d.kT * d.kH * d.kW) + d.sc_Cin)
    rec.append((ops.conv_kernel_name(d), (d.Ti, d.Hi, d.Wi, d.Cin), (d.To, d.Ho, d.Wo, d.Cout), fl, e0, e1))


for it in range(2):
    rec.clear()
    ops.PROFILE = obs if it == 1 else None
    z = vae.encode(x).latent_dist.mode()
    n_enc = len(rec)
    y = vae.decode(z).sample
    torch.cuda.synchronize()
ops.PROFILE = None
tot = 0.0
for i, (name, ishp, oshp, fl, e0, e1) in enumerate(rec):
    ms = e0.elapsed_time(e1)
    tot += ms
    print(f"{'enc' if i < n_enc else 'dec'} {i:3d} {name:48s} in {str(ishp):22s} out {str(oshp):22s} {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
z = vae.encode(x).latent_dist.mode()
y = vae.decode(z).sample
e1.record()
torch.cuda.synchronize()
print(f"sum of conv launches {tot:.2f} ms; whole step (no per-launch events) {e0.elapsed_time(e1):.2f} ms")
