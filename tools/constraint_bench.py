#!/usr/bin/env python
"""Timing of the frozen 2-D constraint decoder (cvvae_amd/constraint.py) at the training config's shapes
(configs/cvvae_sd3_constraint_training.yaml: 17-frame 256x256 clips -> latents [1,16,5,32,32]; 320x320 images, batch 8 ->
[8,16,40,40]) and at a 512x512 clip's latents.  --grad also times forward + input-gradient backward under torch.autograd
(the frozen decoder inside a training step, cvvae_amd/grad.py).  usage: python tools/constraint_bench.py [--iters N] [--hip-graphs] [--grad]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402  (conv_flops: algorithmic FLOPs of a launch)
from cvvae_amd import ops  # noqa: E402
from cvvae_amd.constraint import DecoderWith3DWrapper  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--hip-graphs", action="store_true")
    ap.add_argument("--grad", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    m = DecoderWith3DWrapper(in_channels=16, out_channels=3, up_block_types=["UpDecoderBlock2D"] * 4,
                             block_out_channels=[128, 256, 512, 512], layers_per_block=2).to(torch.bfloat16).cuda().eval()
    m.requires_grad_(False)  # lvdm/models/autoencoder.py:1057-1058
    if a.hip_graphs:
        m.enable_hip_graphs(True)
    for shape in [(1, 16, 5, 32, 32), (8, 16, 40, 40), (1, 16, 5, 64, 64)]:
        z = (torch.rand(shape, device="cuda") * 2 - 1).to(torch.bfloat16)
        fl = [0.0]
        ops.PROFILE = lambda d, pw, launch: (fl.__setitem__(0, fl[0] + bench.conv_flops(d, pw)), launch())
        m.enable_hip_graphs(False)
        y = m(z)
        ops.PROFILE = None
        if a.hip_graphs:
            m.enable_hip_graphs(True)
        for _ in range(3):
            m(z)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            m(z)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        frames = shape[0] * (shape[2] if len(shape) == 5 else 1)
        print(f"latents {shape} -> {tuple(y.shape)}: {ms:7.3f} ms  {frames / ms * 1e3:8.1f} frames/s  "
              f"{fl[0] / ms / 1e9:7.1f} TFLOP/s algorithmic ({fl[0] / 1e12:.3f} TFLOP)", flush=True)
        if a.grad:
            zg = z.clone().requires_grad_(True)
            cot = torch.randn_like(y)
            for _ in range(2):
                zg.grad = None
                m(zg).backward(cot)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.iters):
                zg.grad = None
                m(zg).backward(cot)
            e1.record()
            torch.cuda.synchronize()
            msg = e0.elapsed_time(e1) / a.iters
            print(f"    forward + input-gradient backward: {msg:7.3f} ms ({msg / ms:.2f}x the forward)", flush=True)


if __name__ == "__main__":
    main()
