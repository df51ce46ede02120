#!/usr/bin/env python
"""The ORACLE (oracle/cvvae_oracle.py: plain PyTorch ops, the reference's arithmetic) run on the GPU through PyTorch-ROCm's own
kernels (MIOpen convolutions, native GroupNorm / SiLU / SDPA) -- "what the reference's PyTorch path costs on this MI355X"
(GPU-side second oracle) -- timed beside the HIP path on the same seeded weights and input, and compared with it.
A measurement aid only: nothing in the product imports this.   usage (GPU box): python tools/torch_gpu_baseline.py [--dtype bf16]
"""
import argparse, json, os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "2")  # "fast" find: the exhaustive first-call search of the 3-D convs takes > 10 minutes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import cvvae_oracle as O
from oracle.seeded import seeded_input, seeded_state_dict
from oracle.shapes import state_dict_shapes

ap = argparse.ArgumentParser()
ap.add_argument("--family", default="sd3")
ap.add_argument("--shape", default="1,3,17,512,512")
ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--no-miopen", action="store_true", help="torch.backends.cudnn.enabled = False: ATen's vol2col + rocBLAS GEMM convolutions")
ap.add_argument("--trace-first", action="store_true", help="print the time of every convolution call of the first (warm-up) step to stderr")
a = ap.parse_args()
if a.no_miopen:
    torch.backends.cudnn.enabled = False
if a.trace_first:
    import torch.nn.functional as F
    _t_begin = time.time()

    def _traced(fn, name):
        def f(x, w, *args, **kw):
            torch.cuda.synchronize()
            t0 = time.time()
            y = fn(x, w, *args, **kw)
            torch.cuda.synchronize()
            if _traced.on:
                print(f"[{time.time() - _t_begin:7.1f}s] {name} x{tuple(x.shape)} w{tuple(w.shape)} {1e3 * (time.time() - t0):9.2f} ms", file=sys.stderr, flush=True)
            return y
        return f
    _traced.on = True
    F.conv3d, F.conv2d = _traced(F.conv3d, "conv3d"), _traced(F.conv2d, "conv2d")
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
shape = tuple(int(v) for v in a.shape.split(","))
sd = {k: v.to(dt).cuda() for k, v in seeded_state_dict(state_dict_shapes(a.family), 0).items()}
x = seeded_input(shape, 0).to(dt).cuda()


def step():
    mom = O.encode_moments(x, sd, {}, a.family)
    return mom, O.decode_sample(O.posterior_mode(mom), sd, {}, a.family)


with torch.no_grad():
    t0 = time.time()
    mom, rec = step()  # warm-up (MIOpen picks its kernels here)
    torch.cuda.synchronize()
    warm = time.time() - t0
    if a.trace_first:
        _traced.on = False
    ts = []
    for _ in range(a.iters):
        torch.cuda.synchronize()
        t0 = time.time()
        mom, rec = step()
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
ms = sorted(ts)[len(ts) // 2] * 1e3
out = {"what": "oracle restatement on PyTorch-ROCm (" + ("ATen vol2col + rocBLAS convolutions, cudnn/MIOpen disabled" if a.no_miopen else "MIOpen convolutions") +
       "; ATen GroupNorm / SiLU / SDPA), same seeded weights and input", "family": a.family,
       "shape": list(shape), "dtype": a.dtype, "ms_per_clip": round(ms, 2), "frames_per_s": round(shape[0] * shape[2] / ms * 1e3, 2),
       "first_call_s": round(warm, 1), "torch": torch.__version__, "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1)}

# the HIP path on the same input: time and difference
import cvvae_amd
cls = cvvae_amd.CVVAESD3Model if a.family == "sd3" else cvvae_amd.CVVAEModel
vae = cls()
vae.load_state_dict({k: v.float().cpu() for k, v in sd.items()}, strict=True)
vae = vae.to(dt).cuda().eval()
with torch.no_grad():
    for _ in range(2):
        z = vae.encode(x).latent_dist
        y = vae.decode(z.mode()).sample
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.iters):
        z = vae.encode(x).latent_dist
        y = vae.decode(z.mode()).sample
    torch.cuda.synchronize()
    hip_ms = (time.time() - t0) / a.iters * 1e3
zc = mom.shape[1] // 2
out["hip_ms_per_clip"] = round(hip_ms, 2)
out["speedup"] = round(ms / hip_ms, 2)
out["latent_max_abs_diff"] = float((z.mean.float() - mom[:, :zc].float()).abs().max())
out["recon_max_abs_diff"] = float((y.float() - rec.float()).abs().max())
print(json.dumps(out))
