#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: encode+decode frames/sec of the CV-VAE codec at T=17, 512x512.

One "step" = one pass of the hot path over one clip: `vae.encode(x).latent_dist.mode()` then `vae.decode(z).sample`
for a synthetic [1,3,17,512,512] clip (config.workload = BASELINE cfg 3: vae3d_sd3, bf16).  Inputs are resident in
HBM before the timed region.  N>1 (launched by torch.distributed.run): one process per GPU, every rank codes its own
clip (the path partitions into independent 17-frame windows -- SURVEY.md 8e -- so there is no data-path collective;
weak scaling); value = clips of all ranks * 17 frames / max-over-ranks time.

Besides the contract line, the JSON carries
  roofline     -- the dominant kernel's algorithmic FLOPs / its HIP-event time, measured live in a separate pass
  cpu_baseline -- the CPU oracle (oracle/cvvae_oracle.py, a PyTorch-CPU restatement of the reference: "port") on a
                  bounded sample of the same workload, on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA peak of MI355X (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (family, B, T, H, W)            -- BASELINE.json configs; cfg 3 is the one the metric is quoted on (default)
    "cfg3_sd3_T17_512": ("sd3", 1, 17, 512, 512),
    "cfg2_vae3d_T17_256": ("vae3d", 1, 17, 256, 256),
    # cfg 1: image mode (T = 1) -- the reference's own CPU-runnable case; launch-bound on a GPU (see --hip-graphs)
    "cfg1_vae3d_T1_256": ("vae3d", 1, 1, 256, 256),
    # cfg 4: ONE long clip, its 8 temporal windows (x 6 spatial tiles each) sharded over the ranks (cv-vae_amd/dist.py):
    # strong scaling, encode + decode of the whole clip, gathered latents
    "cfg4_sd3_T129_720x1280": ("sd3", 1, 129, 720, 1280),
    # cfg 5: batch-8 T=33 encode-only (training-side latent pre-compute); the batch is split over the ranks
    "cfg5_sd3_B8_T33_512_encode": ("sd3", 8, 33, 512, 512),
}
ENC_TFLOP = {"cfg3_sd3_T17_512": 22.842, "cfg2_vae3d_T17_256": 5.674, "cfg1_vae3d_T1_256": 0.533}  # encoder share of ALG_TFLOP
# algorithmic FLOPs per unit of work (BASELINE.md section 3 / SURVEY 8d: 2*M*N*K of every conv/linear + attention)
ALG_TFLOP = {"cfg3_sd3_T17_512": 91.41, "cfg2_vae3d_T17_256": 22.79, "cfg4_sd3_T129_720x1280": 3656.0,
             "cfg5_sd3_B8_T33_512_encode": 365.4, "cfg1_vae3d_T1_256": 2.27}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3_sd3_T17_512", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hip-graphs", action="store_true",
                    help="replay each encoder/decoder pass as a captured hipGraph (vae.enable_hip_graphs(); for launch-bound inputs)")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def conv_flops(d, pw):
    # ALGORITHMIC work of the reference op: folded forms count the taps of the op they replace; + the fused 1x1 shortcut
    taps = getattr(pw, "alg_taps", 0) or d.kT * d.kH * d.kW
    return 2.0 * d.B * d.To * d.Ho * d.Wo * d.Cout * (pw.cin_real * taps + d.sc_Cin)


def roofline_pass(step_fn):
    """Re-run one step with a HIP event pair around every conv launch (on the stream it is launched on) and
    aggregate per kernel instance."""
    from cvvae_amd import ops

    rec = []

    def observer(d, pw, launch):
        name = ops.conv_kernel_name(d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        rec.append((name, conv_flops(d, pw), e0, e1))

    ops.PROFILE = observer
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        ops.PROFILE = None
    agg = {}
    for name, fl, e0, e1 in rec:
        a = agg.setdefault(name, [0.0, 0.0, 0])
        a[0] += fl
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
    return agg


def cpu_baseline(family, H=512, W=512):
    """CPU oracle on a bounded sample: the same network on a 17-frame window at 192x192 (0.14 of the 512x512 frame
    area; one temporal window, no spatial tiling -- like the workload), scaled by pixel count to the workload's frame size."""
    from oracle import cvvae_oracle as O
    from oracle.seeded import seeded_input, seeded_state_dict
    from oracle.shapes import state_dict_shapes

    cores = min(os.cpu_count() or 1, 32)  # oneDNN conv at this size stops scaling (and regresses) beyond ~32 threads
    torch.set_num_threads(cores)
    sd = seeded_state_dict(state_dict_shapes(family), 0)
    hw = 192  # ~10 s of CPU work on the GPU box's host cores
    x = seeded_input((1, 3, 17, hw, hw), 0)
    with torch.no_grad():
        t0 = time.time()
        mom = O.encode_moments(x, sd, {}, family)
        rec = O.decode_sample(O.posterior_mode(mom), sd, {}, family)
        dt = time.time() - t0
    assert rec.shape == x.shape
    area_scale = (H * W) / float(hw * hw)
    return {
        "value": round(17.0 / (dt * area_scale), 5),
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "sample": f"oracle (PyTorch-CPU fp32 restatement) encode+decode of 1x3x17x{hw}x{hw} in {dt:.1f}s; value = 17 frames "
                  f"/ (t * {area_scale:.1f}) i.e. scaled by pixel count to the {H}x{W} workload",
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_

        dist = dist_
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import cvvae_amd

    family, B, T, H, W = WORKLOADS[args.workload]
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    torch.manual_seed(0)
    cls = cvvae_amd.CVVAESD3Model if family == "sd3" else cvvae_amd.CVVAEModel
    vae = cls().to(dtype).cuda().eval()  # random-init weights of the named architecture (no checkpoint access)
    if args.hip_graphs:
        vae.enable_hip_graphs(True)
    cfg4 = args.workload.startswith("cfg4")
    cfg5 = args.workload.startswith("cfg5")
    if cfg5:
        if B % world:
            raise SystemExit(f"cfg5 splits its batch of {B} over the ranks: --gpus must divide {B}")
        B = B // world
    # cfg 3 / 2: every rank codes its own clip (weak scaling).  cfg 4: every rank holds the same clip (seed without rank)
    g = torch.Generator().manual_seed(1000 + (0 if cfg4 else rank))
    x = (torch.rand((B, 3, T, H, W), generator=g) * 2 - 1).to(dtype).cuda()

    if cfg4 and dist is not None:
        from cvvae_amd import dist as D

        def step():
            mom = D.encode_windows_sharded(vae, x)                     # all_gather of the latents (15 MB)
            z = mom[:, :mom.shape[1] // 2]
            return D.decode_windows_sharded(vae, z, gather=False)      # pixels stay sharded (713 MB if gathered)
    elif cfg5:
        def step():
            return vae.encode(x).latent_dist.mode()
    else:
        def step():
            z = vae.encode(x).latent_dist.mode()
            return vae.decode(z).sample

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert cfg5 or (cfg4 and dist is not None) or y.shape == x.shape
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    strong = cfg4                                 # one fixed clip split over the ranks; everything else: per-rank work fixed
    frames = (1 if strong else world) * B * T * args.steps
    units = (1 if strong else world) * (B if cfg5 else 1) * args.steps / (8 if cfg5 else 1)  # ALG_TFLOP units done
    out = {
        "metric": "encode+decode frames/sec (T=17, 512x512)" if args.workload.startswith("cfg3") else
                  ("encode-only frames/sec" if cfg5 else "encode+decode frames/sec"),
        "value": round(frames / elapsed, 3),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic uniform[-1,1) clip, random-init weights (seed 0) of the named architecture",
        "config": {"workload": f"{args.workload}: {family} " + ("encode(x).mode()" if cfg5 else "encode(x).mode() + decode(z)") +
                               f", x=[{B},3,{T},{H},{W}] per GPU",
                   "clips_per_gpu": B, "hip_graphs": bool(args.hip_graphs),
                   "parallelism": (f"temporal windows sharded x{world}, latents all-gathered" if strong else
                                   f"independent clips x{world} (no collective)")},
        "achieved_tflops_whole_path": round(ALG_TFLOP[args.workload] * units / elapsed, 1),
    }

    if rank == 0 and not (cfg5 or (cfg4 and dist is not None)):
        # encode / decode split of one step (separate, untimed pass; events on the launch stream)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for _ in range(2):
            ev[0].record()
            zz = vae.encode(x).latent_dist.mode()
            ev[1].record()
            vae.decode(zz)
            ev[2].record()
        torch.cuda.synchronize()
        enc_ms, dec_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
        out["encode_ms"], out["decode_ms"] = round(enc_ms, 3), round(dec_ms, 3)
        if args.workload in ENC_TFLOP:
            et = ENC_TFLOP[args.workload]
            out["encode_tflops"] = round(et / enc_ms * 1e3, 1)
            out["decode_tflops"] = round((ALG_TFLOP[args.workload] - et) / dec_ms * 1e3, 1)
            out["encode_frac_of_mfma_peak"] = round(et / enc_ms * 1e3 / MFMA_PEAK_TFLOPS, 4)
    if rank == 0 and not args.no_roofline:
        vae.enable_hip_graphs(False)  # per-launch timing needs the eager launches
        agg = roofline_pass(step)
        name, (fl, sec, n) = max(agg.items(), key=lambda kv: kv[1][1])
        ach = fl / sec / 1e12
        # HBM traffic per launch of that kernel: from the committed PMC passes of this same command (profiles/)
        traffic = None
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.isfile(tj):
            with open(tj) as f:
                k = json.load(f).get("kernels", {}).get(name.split("_", 3)[3] if name.count("_") >= 3 else name)
            if k:
                traffic = k["fetch_bytes"] + k["write_bytes"]
        out["roofline"] = {
            "bound": "mfma", "kernel": name, "launches_per_step": n,
            "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
            "avg_launch_ms": round(sec / n * 1e3, 4), "alg_gflop_per_launch": round(fl / n / 1e9, 2), "traffic": traffic,
            "traffic_unit": "bytes/launch (PMC, profiles/pmc_traffic.json)",
        }
        out["kernels"] = {k: {"tflops": round(v[0] / v[1] / 1e12, 1), "ms": round(v[1] * 1e3, 3), "launches": v[2]}
                          for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(family, H, W)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
