#!/usr/bin/env python
"""Micro-benchmark of single conv launches (tuning aid; also the command profiled for profiles/*pmc*).
usage: python tools/conv_bench.py [case ...] [--iters N] [--dtype bf16|f16|f32|f32q|f32q6|f32ab]
(f32 = three fp16 MFMAs per product, f32q = fp16 + bf8 corrections, f32q6 = fp16 + fp6 corrections where an instance exists,
f32ab = the last two interleaved: labels +q / +q6)
cases are the cfg-3 layer shapes of SURVEY.md 2.3."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cvvae_amd  # noqa: E402,F401
from cvvae_amd import _lib as L  # noqa: E402
from cvvae_amd import ops  # noqa: E402

P1 = ((1, 1), (1, 1), (1, 1))
PC = ((2, 0), (1, 1), (1, 1))
P2D = ((0, 0), (1, 1), (1, 1))
# name: (Cin, Cout, k, T, H, W, pad, prologue, ups)
CASES = {
    "enc128": (128, 128, (3, 3, 3), 17, 512, 512, PC, 1, False),      # 128->128 @17x512^2 (config B)
    "enc256": (256, 256, (3, 3, 3), 9, 256, 256, PC, 1, False),       # 256->256 @9x256^2 (config A)
    "enc512": (512, 512, (3, 3, 3), 9, 128, 128, PC, 1, False),
    "mid512_64": (512, 512, (3, 3, 3), 9, 64, 64, P1, 1, False),      # vae3d cfg 2: 288 workgroups with 256-pixel tiles
    "dec512s": (512, 512, (3, 3, 3), 5, 64, 64, P1, 1, False),
    "dec256to128": (256, 128, (3, 3, 3), 17, 512, 512, P1, 1, False),
    "up256to512": (256, 512, (3, 3, 3), 9, 256, 256, P1, 0, True),    # the 16.7 TFLOP upsample conv
    "upfold256to512": (256, 512, (3, 3, 3), 9, 256, 256, P1, 0, 2),  # the same conv as four folded 3x2x2 phase convs
    "upfold512": (512, 512, (3, 3, 3), 9, 128, 128, P1, 0, 2),
    "down128": (128, 128, (3, 3, 3), 17, 512, 512, PC, 0, False),     # encoder downsamplers (stride set below)
    "down256": (256, 256, (3, 3, 3), 9, 256, 256, PC, 0, False),
    "down512": (512, 512, (3, 3, 3), 9, 128, 128, PC, 0, False),
    "c2d128": (128, 128, (1, 3, 3), 17, 512, 512, P2D, 1, False),
    "c2d512": (512, 512, (1, 3, 3), 9, 128, 128, P2D, 1, False),
    "out128to3": (128, 3, (3, 3, 3), 17, 512, 512, P1, 1, False),
    # ResnetBlock tails: per-frame conv2 + residual add + fused GroupNorm statistics of the sum (name suffix "res")
    # boundary-frame probes for the time folds (--tfolds): T = 1 causal = every tile multiplies ONE frame (W0+W1+W2), T = 2 adds
    # the two-slot frame; the ratio to the plain weights is the realised cost of such tiles
    "enc256T1": (256, 256, (3, 3, 3), 1, 256, 256, PC, 1, False),
    "enc256T2": (256, 256, (3, 3, 3), 2, 256, 256, PC, 1, False),
    "enc128T2": (128, 128, (3, 3, 3), 2, 512, 512, PC, 1, False),
    "upfoldT2": (256, 512, (3, 3, 3), 2, 256, 256, P1, 0, 2),
    "enc256q": (256, 256, (3, 3, 3), 9, 248, 256, PC, 1, False),   # 2232 workgroups = 8.72 rounds (enc256: exactly 9)
    "enc256r": (256, 256, (3, 3, 3), 9, 264, 256, PC, 1, False),   # 2376 workgroups = 9.28 rounds
    "c2d128res": (128, 128, (1, 3, 3), 17, 512, 512, P2D, 1, False),
    "c2d256res": (256, 256, (1, 3, 3), 9, 256, 256, P2D, 1, False),
    "c2d512res": (512, 512, (1, 3, 3), 9, 128, 128, P2D, 1, False),
    # the small-frame 512-channel layers of cfg 1 / cfg 2 (17x256^2 and 1x256^2 clips): few workgroups, 14 MB of weights per layer
    "mid512_32": (512, 512, (3, 3, 3), 5, 32, 32, P1, 1, False),
    "c2d512_32res": (512, 512, (1, 3, 3), 5, 32, 32, P2D, 1, False),
    "c2d512_32T1res": (512, 512, (1, 3, 3), 1, 32, 32, P2D, 1, False),
    "c2d512_64T1res": (512, 512, (1, 3, 3), 1, 64, 64, P2D, 1, False),
    "c2d512_64res": (512, 512, (1, 3, 3), 9, 64, 64, P2D, 1, False),    # cfg 2: 288 workgroups of 256 x 256 = 1.125 rounds
    "c2d256_128res": (256, 256, (1, 3, 3), 9, 128, 128, P2D, 1, False),
    "mid256_128": (256, 256, (3, 3, 3), 9, 128, 128, P1, 1, False),
    "c2d256_128T1res": (256, 256, (1, 3, 3), 1, 128, 128, P2D, 1, False),   # cfg 1 (image mode at 256^2): its three upper levels
    "c2d128_256T1res": (128, 128, (1, 3, 3), 1, 256, 256, P2D, 1, False),
}


def setenv(f):
    """f = "FORCE" or "FORCE@ORDER" (CVVAE_CONV_FORCE / CVVAE_CONV_ORDER tuning knobs of libcvvae_hip.so)"""
    f, _, envs = f.partition("!")  # "...!NAME=VAL,NAME=VAL": extra library knobs for this variant (reset otherwise)
    for k in ("CVVAE_CONV_PHASE_SYNC",):
        os.environ.pop(k, None)
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        os.environ[k] = v
    force, _, order = f.partition("@")
    os.environ["CVVAE_CONV_FORCE"] = force
    if order:
        os.environ["CVVAE_CONV_ORDER"] = order
    else:
        os.environ.pop("CVVAE_CONV_ORDER", None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=["enc128", "enc256", "up256to512", "c2d128"])
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--force", action="append", default=None,
                    help='CVVAE_CONV_FORCE values to A/B ("" = library default), e.g. --force "" --force 1x8x32:2x4x1:1')
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--tfolds", action="store_true", help="also time every 3x3x3 case with the time-fold weight slots (force label +tf)")
    a = ap.parse_args()
    forces = a.force or [""]
    dt = torch.bfloat16 if a.dtype == "bf16" else (torch.float16 if a.dtype == "f16" else torch.float32)
    fasts = {"f32": [("", False)], "f32q": [("", True)], "f32q6": [("", "fp6")], "f32ab": [("+q", True), ("+q6", "fp6")]}.get(a.dtype, [("", False)])
    for name in a.cases:
        cin, cout, k, T, H, W, pad, pro, ups = CASES[name]
        x = (torch.rand((1, T, H, W, cin), device="cuda") * 2 - 1).to(dt)
        w = (torch.rand((cout, cin) + k, device="cuda") * 2 - 1).to(dt) / (cin * k[0] * k[1] * k[2]) ** 0.5
        def packed(fast, tf=False):
            fast = True if (fast == "fp6" and not ((pro == 1 and not ups and k in ((3, 3, 3), (1, 3, 3)) and not name.startswith("down")) or ups == 2)) else fast
            if ups == 2:
                q = ops.pack_weight_upfold(w, torch.zeros(cout, device="cuda"), time_folds=tf, fast=fast)
            elif tf:
                q = ops.pack_weight_tfolds(w, torch.zeros(cout, device="cuda"), fast=fast)
            else:
                q = ops.pack_weight(w.reshape(cout, cin, -1), torch.zeros(cout, device="cuda"), k, fast=fast)
            if q.dt == L.F32Q6 and pro:
                q.act_bound = 8.0
            return q
        pw = packed(fasts[0][1])
        gn = None
        if pro:
            gn = ops.gn_stats(x, torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda"), 1e-6)
        if name.startswith("down"):
            stride = (1, 2, 2) if name == "down256" else (2, 2, 2)
        else:
            stride = (1, 1, 1)
        kw = dict(stride=stride, pad=pad, pad_mode_t=L.PAD_REPLICATE, pad_mode_hw=L.PAD_REPLICATE if k[0] == 3 else L.PAD_ZERO, prologue=pro,
                  gn=gn, upsample2x=ups, out_mode=L.OUT_NCDHW if cout <= 32 else L.OUT_NDHWC)
        if pw.dt == L.F32Q6 and not pro:  # fp6 without a GroupNorm in front (folded upsample): the bound lives on the device
            kw["act_bound_dev"] = torch.linalg.vector_norm(x.reshape(-1), float("inf")).reshape(1)
        if name.endswith("res"):
            kw["residual"] = (torch.rand((1, T, H, W, cout), device="cuda") * 2 - 1).to(dt)
            kw["gn_out"] = 32
        npix = None
        best, kname = {}, {}
        pws = {f + lab: packed(fast) for f in forces for lab, fast in fasts}
        if a.tfolds and k[0] == 3:
            pws.update({f + lab + "+tf": packed(fast, True) for f in forces for lab, fast in fasts})
        forces_c = list(pws)
        for rnd in range(a.rounds):  # interleaved rounds: A/B deltas come from one process (guide rule 24)
            for f in forces_c:
                pw = pws[f]
                setenv(f.replace("+tf", "").replace("+q6", "").replace("+q", ""))  # (labels: FORCE[@ORDER][!ENV=VAL,...][+q|+q6][+tf])
                ops.PROFILE = lambda d, pw_, launch, f=f: (kname.__setitem__(f, ops.conv_kernel_name(d)), launch())
                y = ops.conv(x, pw, **kw)
                if isinstance(y, tuple):
                    y = y[0]
                ops.PROFILE = None
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    ops.conv(x, pw, out=y, **kw)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.iters
                best.setdefault(f, []).append(ms)
                npix = y.numel() // cout
        fl = 2.0 * npix * cout * cin * k[0] * k[1] * k[2]
        for f in forces_c:
            ms = sorted(best[f])[len(best[f]) // 2]
            print(f"{name:12s} force={f or '-':22s} median {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s  min {min(best[f]):8.3f} ms "
                  f"({fl / 1e9:.0f} GFLOP)  {kname.get(f, '')}", flush=True)
    setenv("")


if __name__ == "__main__":
    main()
