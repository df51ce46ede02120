#!/usr/bin/env python
"""For every distinct convolution of one encode+decode step, time EVERY kernel instance that can run it (CVVAE_CONV_FORCE)
against the library's own choice (interleaved rounds, one process).  Shows where the cost model of select_instance() leaves
performance on the table.  usage (GPU box): python tools/tune_instances.py [--family sd3|vae3d] [--shape 1,3,17,512,512]"""
import argparse, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CVVAE_CONV_TUNE_NOSTATS"] = "1"  # forced instances must not write into record tables sized for the default one
import torch
import cvvae_amd
from cvvae_amd import ops, _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--family", default="sd3")
ap.add_argument("--shape", default="1,3,17,512,512")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--train", action="store_true", help="sweep the convolutions of one TRAINING step (taped forward + backward: the input-gradient convs too)")
a = ap.parse_args()
torch.manual_seed(0)
dtype = torch.bfloat16
vae = (cvvae_amd.CVVAESD3Model if a.family == "sd3" else cvvae_amd.CVVAEModel)().to(dtype).cuda()
vae = vae.train() if a.train else vae.eval()
x = (torch.rand(tuple(int(v) for v in a.shape.split(","))) * 2 - 1).to(dtype).cuda()

# every X(...) row of the instance table -> a force string
rows = re.findall(r"X\(([^)]*)\)", open(os.path.join(ROOT, "cvvae_amd", "csrc", "conv_table.h")).read())
forces = sorted({"%sx%sx%s:%sx%sx%s:%s" % tuple(r.replace(" ", "").split(",")[i] for i in (6, 7, 8, 9, 10, 11, 12)) for r in rows if r[0].isdigit()})

calls = {}

def obs(d, pw, launch):
    key = (d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.Ti, d.Hi, d.Wi, d.Cin, d.Cout, d.prologue, d.upsample2x, d.out_mode, d.sc_Cin)
    if key not in calls:
        calls[key] = [launch, ops.conv_kernel_name(d), 0, d]
    calls[key][2] += 1
    launch()

ops.PROFILE = obs
if a.train:
    mom = vae.encoder(x)
    xrec = vae.decoder(mom[:, :mom.shape[1] // 2].contiguous())
    (xrec.float() - x.float()).pow(2).mean().backward()
else:
    z = vae.encode(x).latent_dist.mode()
    y = vae.decode(z).sample
torch.cuda.synchronize()
ops.PROFILE = None

def timeit(launch, n=3):
    launch(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

lib = L.load()
total_def = total_best = 0.0
for key, (launch, defname, count, d) in sorted(calls.items(), key=lambda kv: -kv[1][2]):
    cands = {"": defname}
    for f in forces:
        os.environ["CVVAE_CONV_FORCE"] = f
        n = ops.conv_kernel_name(d)
        if n and n != defname and n not in cands.values():
            cands[f] = n
    os.environ["CVVAE_CONV_FORCE"] = ""
    if len(cands) == 1:
        t = timeit(launch)
        total_def += t * count; total_best += t * count
        continue
    res = {f: [] for f in cands}
    for _ in range(a.rounds):
        for f in cands:
            os.environ["CVVAE_CONV_FORCE"] = f
            res[f].append(timeit(launch))
    os.environ["CVVAE_CONV_FORCE"] = ""
    med = {f: sorted(v)[len(v) // 2] for f, v in res.items()}
    best = min(med, key=med.get)
    total_def += med[""] * count; total_best += med[best] * count
    flag = "" if best == "" or med[best] > 0.97 * med[""] else "   <-- %.1f%% faster" % ((med[""] / med[best] - 1) * 100)
    print(f"k{key[0]}{key[1]}{key[2]} s{key[3]}{key[4]}{key[5]} in {key[6]}x{key[7]}x{key[8]}x{key[9]} -> {key[10]} pro{key[11]} ups{key[12]} x{count}: "
          f"default {cands['']} {med['']:.3f} ms{flag}")
    for f in cands:
        if f:
            print(f"      {cands[f]:52s} {med[f]:.3f} ms")
print(f"sum over the step: default choices {total_def:.2f} ms, best measured {total_best:.2f} ms")
