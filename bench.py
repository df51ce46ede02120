#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: encode+decode frames/sec of the CV-VAE codec at T=17, 512x512.

One "step" = one pass of the hot path over one clip: `vae.encode(x).latent_dist.mode()` then `vae.decode(z).sample`
for a synthetic [1,3,17,512,512] clip (config.workload = BASELINE cfg 3: vae3d_sd3, bf16).  Inputs are resident in
HBM before the timed region.

N > 1: one process per GPU over RCCL.  `python bench.py --gpus N` launches the N ranks itself (it re-executes under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`); started by an external launcher
(RANK / WORLD_SIZE in the environment) it simply is one of the ranks.  cfg 3 / 2 / 5: every rank codes its own clip(s)
(the path partitions into independent 17-frame windows -- SURVEY.md 8e -- so there is no data-path collective; weak
scaling); cfg 4: ONE clip, its temporal windows sharded over the ranks (cvvae_amd/dist.py; strong scaling).
value = frames of all ranks / max-over-ranks time.

Besides the contract line, the JSON carries
  roofline     -- the dominant kernel's algorithmic FLOPs / its HIP-event time, measured live in a separate pass
                  (+ the EXECUTED MFMA FLOPs of the same launches: the algebraic folds remove work)
  parity       -- latent max / mean |delta| and recon PSNR of THIS build against the reference-generated fixture of the
                  workload (tests/golden/cfg3_sd3_t17_512.npz), in the bench dtype, next to the reference's own noise
  hbm          -- algorithmic GB/s of the step and the PMC-counted traffic of the committed profile (profiles/)
  mfma_busy    -- SQ counter ratio of the dominant kernel from the committed PMC pass (profiles/pmc_sq.json)
  cpu_baseline -- the CPU oracle (oracle/cvvae_oracle.py, a PyTorch-CPU restatement of the reference: "port") on a
                  bounded sample of the same workload, on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA peak of MI355X (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0     # HBM3E, same guide

WORKLOADS = {
    # name: (family, B, T, H, W)            -- BASELINE.json configs; cfg 3 is the one the metric is quoted on (default)
    "cfg3_sd3_T17_512": ("sd3", 1, 17, 512, 512),
    "cfg2_vae3d_T17_256": ("vae3d", 1, 17, 256, 256),
    # cfg 1: image mode (T = 1) -- the reference's own CPU-runnable case; launch-bound on a GPU (see --hip-graphs)
    "cfg1_vae3d_T1_256": ("vae3d", 1, 1, 256, 256),
    # cfg 4: ONE long clip, its 8 temporal windows (x 6 spatial tiles each) sharded over the ranks (cvvae_amd/dist.py):
    # strong scaling, encode + decode of the whole clip, gathered latents
    "cfg4_sd3_T129_720x1280": ("sd3", 1, 129, 720, 1280),
    # cfg 5: batch-8 T=33 encode-only (training-side latent pre-compute); the batch is split over the ranks
    "cfg5_sd3_B8_T33_512_encode": ("sd3", 8, 33, 512, 512),
}
ENC_TFLOP = {"cfg3_sd3_T17_512": 22.842, "cfg2_vae3d_T17_256": 5.674, "cfg1_vae3d_T1_256": 0.533}  # encoder share of ALG_TFLOP
# algorithmic FLOPs per unit of work (BASELINE.md section 3 / SURVEY 8d: 2*M*N*K of every conv/linear + attention)
ALG_TFLOP = {"cfg3_sd3_T17_512": 91.41, "cfg2_vae3d_T17_256": 22.79, "cfg4_sd3_T129_720x1280": 3656.0,
             "cfg5_sd3_B8_T33_512_encode": 365.4, "cfg1_vae3d_T1_256": 2.27}
# algorithmic HBM bytes per unit (every conv/linear reads its input once, writes its output once)
ALG_GB = {"cfg3_sd3_T17_512": 48.9, "cfg2_vae3d_T17_256": 12.7, "cfg4_sd3_T129_720x1280": 1945.0,
          "cfg5_sd3_B8_T33_512_encode": 268.0, "cfg1_vae3d_T1_256": 1.5}
GOLDEN_OF = {"cfg3_sd3_T17_512": "cfg3_sd3_t17_512", "cfg2_vae3d_T17_256": "cfg2_vae3d_t17_256",
             "cfg1_vae3d_T1_256": "cfg1_vae3d_t1_256"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3_sd3_T17_512", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32", "f32q"],
                    help="model dtype; f32 = fp32 model in split precision (three fp16 MFMAs per product), f32q = fp32 model in fast "
                         "split precision (fp16 MFMA + bf8 / fp6 correction MFMA): the cheapest mode inside north_star's 1e-3 latent bound")
    ap.add_argument("--no-tolerance-mode", action="store_true",
                    help="skip the annex that times and checks the f32q model (the mode that meets |delta| <= 1e-3) beside the bench dtype")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="time the oracle on the FULL workload shape with all host cores (minutes) instead of the bounded sample")
    ap.add_argument("--hip-graphs", action="store_true",
                    help="replay each encoder/decoder pass as a captured hipGraph (vae.enable_hip_graphs(); for launch-bound inputs)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="cfg 4, N > 1: skip the sharded == single-process latent check")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside a launcher: start the N ranks under torch.distributed.run and relay rank 0's line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def time_groups(to, sT, pad_t, Ti, replicate, has_slots):
    """number of time groups the kernel walks for output frame `to` of a 3-tap time kernel (tile_map.h time_fold_plan)"""
    f0 = to * sT - pad_t
    if replicate:
        if not has_slots:
            return 3
        c = [min(max(f0 + i, 0), Ti - 1) for i in range(3)]
        return 1 if c[0] == c[1] == c[2] else (2 if (c[0] == c[1] or c[1] == c[2]) else 3)
    z0, z2 = not (0 <= f0 < Ti), not (0 <= f0 + 2 < Ti)
    return 1 if (z0 and z2) else (2 if (z0 or z2) else 3)


def conv_flops(d, pw):
    # ALGORITHMIC work of the reference op: folded forms count the taps of the op they replace; + the fused 1x1 shortcut
    taps = getattr(pw, "alg_taps", 0) or d.kT * d.kH * d.kW
    return 2.0 * d.B * d.To * d.Ho * d.Wo * d.Cout * (pw.cin_real * taps + d.sc_Cin)


def conv_flops_executed(d, pw):
    """MFMA FLOPs the launch really issues on useful rows/columns: the kernel's own taps (12 per phase pixel for the folded
    upsample, kH*kW for a single-frame fold) over the padded channel counts, minus the time groups the time folds skip."""
    sp = 4 if d.upsample2x == 2 else d.kH * d.kW  # spatial taps per output pixel as executed
    if d.kT == 3:
        rep = d.pad_mode_t == 1
        tg = sum(time_groups(to, d.sT, d.pad_t, d.Ti, rep, bool(d.w_time_folds)) for to in range(d.To))
    else:
        tg = d.To
    return 2.0 * d.B * tg * d.Ho * d.Wo * d.Cout * (d.Cin * sp) + 2.0 * d.B * d.To * d.Ho * d.Wo * d.Cout * d.sc_Cin


def roofline_pass(step_fn):
    """Re-run one step with a HIP event pair around every conv launch (on the stream it is launched on) and
    aggregate per kernel instance."""
    import torch
    from cvvae_amd import ops

    rec = []

    def observer(d, pw, launch):
        name = ops.conv_kernel_name(d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        # (an odd frame count under a two-frame tile is split by the library: frames [0, To-1) on the two-frame-tile kernel, the last
        #  frame on its one-frame-tile twin -- ONE launch here, TWO dispatches in a rocprofv3 trace; cvvae_api.hip odd_frame_sibling)
        split = "_t2x" in name and d.To % 2 == 1 and d.To >= 3 and d.upsample2x != 2 and d.sT == 1
        rec.append((name, conv_flops(d, pw), conv_flops_executed(d, pw), e0, e1, (d.To - 1) / d.To if split else 1.0))

    ops.PROFILE = observer
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        ops.PROFILE = None
    agg = {}
    for name, fl, fx, e0, e1, main_share in rec:
        a = agg.setdefault(name, [0.0, 0.0, 0, 0.0, 0.0])
        a[0] += fl
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
        a[3] += fx
        a[4] += fl * main_share  # algorithmic FLOPs of the main dispatch alone
    return agg


def cpu_baseline(family, T=17, H=512, W=512, full=False):
    """CPU oracle on a bounded sample: the same network on a 17-frame window at 192x192 (0.14 of the 512x512 frame
    area; one temporal window, no spatial tiling -- like the workload), scaled by pixel count to the workload's frame size.
    full=True: the workload's own shape, every host core, no extrapolation."""
    import torch
    from oracle import cvvae_oracle as O
    from oracle.seeded import seeded_input, seeded_state_dict
    from oracle.shapes import state_dict_shapes

    # oneDNN conv at the sample size stops scaling (and regresses) beyond ~32 threads; the full shape takes every core
    cores = (os.cpu_count() or 1) if full else min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = seeded_state_dict(state_dict_shapes(family), 0)
    hh, ww = (H, W) if full else (min(H, 192), min(W, 192))  # sample: ~10 s of CPU work on the GPU box's host cores
    x = seeded_input((1, 3, T, hh, ww), 0)
    with torch.no_grad():
        t0 = time.time()
        mom = O.encode_moments(x, sd, {}, family)
        t1 = time.time()
        rec = O.decode_sample(O.posterior_mode(mom), sd, {}, family)
        dt = time.time() - t0
    assert rec.shape == x.shape
    area_scale = (H * W) / float(hh * ww)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    out = {
        "value": round(T / (dt * area_scale), 5),
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "extrapolated": area_scale != 1.0,
        "cpu_model": cpu_model,
        "encode_s": round(t1 - t0, 2), "decode_s": round(dt - (t1 - t0), 2),
        "sample": (f"oracle (PyTorch-CPU fp32 restatement) encode+decode of the full 1x3x{T}x{hh}x{ww} workload in {dt:.1f}s"
                   if area_scale == 1.0 else
                   f"oracle (PyTorch-CPU fp32 restatement) encode+decode of 1x3x{T}x{hh}x{ww} in {dt:.1f}s; value = {T} frames "
                   f"/ (t * {area_scale:.1f}) i.e. scaled by pixel count to the {H}x{W} workload"),
    }
    # The bounded sample flatters the CPU (oneDNN runs the small frames ~2.5x more efficiently than the 512x512 ones), so the reported
    # `value` is the FULL-size measurement whenever one is kept for this host CPU model (bench.py --cpu-baseline-full, minutes of CPU
    # time, measured once per CPU model and committed as profiles/cpu_baseline_full.json); the in-run sample is the annex that shows
    # the host is the same class of machine.  A host without a kept measurement is measured at full size in this run.
    kept = os.path.join(ROOT, "profiles", "cpu_baseline_full.json")
    if not full:
        k = None
        if os.path.isfile(kept):
            with open(kept) as f:
                k = json.load(f)
        if k is not None and k.get("cpu_model") == cpu_model:
            sample = dict(out)
            out = dict(k, measured_in_this_run=False, in_run_sample=sample,
                       sample_to_full_ratio=round(sample["value"] / k["value"], 2))
        else:
            out = cpu_baseline(family, T, H, W, full=True)
            out["measured_in_this_run"] = True
    return out


def _rccl_version(torch):
    try:  # (diagnostic only: never let it take the bench line down)
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:  # noqa: BLE001
        return f"unavailable ({type(e).__name__})"


def torch_rocm_baseline(family, shape, dtype_tag, ms_per_step):
    """The reference's arithmetic on PyTorch-ROCm's own kernels (MIOpen / ATen) on an MI355X: a KEPT measurement of
    tools/torch_gpu_baseline.py (profiles/r3_torch_gpu_baseline.json; it needs minutes and is not re-run here)."""
    f = os.path.join(ROOT, "profiles", "r3_torch_gpu_baseline.json")
    try:
        runs = json.load(open(f))["runs"]
    except (OSError, ValueError, KeyError):
        return None
    hit = [r for r in runs if r["family"] == family and r["shape"] == list(shape) and r["dtype"] == dtype_tag]
    if not hit:
        return None
    noise = None  # the same ops' own 16-bit noise on the GPU against the reference's fp32 fixtures (tools/torch_gpu_noise.py)
    try:
        allnoise = json.load(open(os.path.join(ROOT, "profiles", "r3_torch_gpu_noise.json")))
        for case, e in allnoise.items():
            if e.get(dtype_tag, {}).get("shape") == list(shape):
                n = e[dtype_tag]
                noise = {"latent_max_abs": n["latent_max_abs"], "latent_mean_abs": n["latent_mean_abs"], "recon_psnr_db": round(n["recon_psnr_db"], 2),
                         "what": "PyTorch-ROCm's own run of these ops in this dtype vs the reference's fp32 CPU fixture (profiles/r3_torch_gpu_noise.json)"}
    except (OSError, ValueError, KeyError):
        pass
    return {"source": "profiles/r3_torch_gpu_baseline.json (tools/torch_gpu_baseline.py, round-3 gpurun box)", "measured_in_this_run": False,
            "own_noise_same_dtype": noise,
            "runs": [{"kernels": "MIOpen" if "MIOpen conv" in r["what"] else "ATen vol2col + rocBLAS", "ms_per_clip": r["ms_per_clip"],
                      "frames_per_s": r["frames_per_s"], "this_run_speedup": round(r["ms_per_clip"] / ms_per_step, 1)} for r in hit]}


def reference_noise(case, dtype_tag):
    from oracle import parity as P
    return P.reference_self_noise(case, dtype_tag)


def profile_json(name):
    p = os.path.join(ROOT, "profiles", name)
    if os.path.isfile(p):
        with open(p) as f:
            return json.load(f)
    return None


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks; running {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_

        dist = dist_
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import cvvae_amd
    from oracle import parity as P  # checker only: seeded weights + the fixture comparison (never inside the timed region)

    family, B, T, H, W = WORKLOADS[args.workload]
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32q": torch.float32}[args.dtype]
    cls = cvvae_amd.CVVAESD3Model if family == "sd3" else cvvae_amd.CVVAEModel
    vae = cls()
    P.load_seeded(vae, 0)  # random weights of the named architecture with PyTorch's default-init statistics (no checkpoint access)
    vae = vae.to(dtype).cuda().eval()
    if dtype == torch.float32:
        vae.fp32_mode = "fast" if args.dtype == "f32q" else "exact"
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
            # (window x spatial tile) network calls split over the ranks (8 x 6 = 48 units at cfg 4: one window per rank on 8 GPUs,
            # no tile traffic; fewer windows than ranks -> the tiles of a window spread out, raw tiles go point-to-point to the
            # rank that blends the window)
            mom = D.encode_units_sharded(vae, x)                       # all_gather of the latents (15 MB)
            z = mom[:, :mom.shape[1] // 2]
            return D.decode_units_sharded(vae, z, gather=False)        # pixels stay sharded (713 MB if gathered)
    elif cfg5:
        def step():
            return vae.encode_latents(x, sample=False)                 # the latent pre-compute API
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
    rank_times = None
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        ts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(ts, t)                      # per-rank clocks (diagnostic); the reported time is their MAX
        rank_times = [round(float(v.item()) / args.steps * 1e3, 3) for v in ts]
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
        "data": "synthetic uniform[-1,1) clip, seeded random weights with default-init statistics of the named architecture",
        "config": {"workload": f"{args.workload}: {family} " + ("encode(x).mode()" if cfg5 else "encode(x).mode() + decode(z)") +
                               f", x=[{B},3,{T},{H},{W}] per GPU",
                   "clips_per_gpu": B, "hip_graphs": bool(args.hip_graphs),
                   "parallelism": (f"(temporal window x spatial tile) network calls sharded x{world}, latents all-gathered" if strong else
                                   f"independent clips x{world} (no collective)")},
        "multi_gpu": None if dist is None else {
            "n_ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
            "rccl_version": _rccl_version(torch),
            "ms_per_step_by_rank": rank_times, "devices": torch.cuda.device_count(),
            "device_name": torch.cuda.get_device_name(local_rank),
            "data_path_collectives": ("all_gather of the latents (15 MB) once per step; tile results point-to-point only when a "
                                      "window's tiles span ranks") if strong else "none (independent clips)",
            "note": "value = work of all ranks / MAX over ranks of the barrier-bracketed time"},
        "achieved_tflops_whole_path": round(ALG_TFLOP[args.workload] * units / elapsed, 1),
        "hbm": {"algorithmic_gbps": round(ALG_GB[args.workload] * units / elapsed, 1), "peak_gbps": HBM_PEAK_GBPS,
                "note": "algorithmic bytes (BASELINE.md section 3) / step time; the path is MFMA-bound (AI ~1900 FLOP/B)"},
    }

    if cfg4 and dist is not None and not args.no_check:
        # the sharded run must reproduce the single-process wrapper bit for bit (windows are independent network calls)
        from cvvae_amd import dist as D
        mom = D.encode_units_sharded(vae, x)
        ok = True
        if rank == 0:
            ref = vae.encode(x).latent_dist.parameters
            ok = bool(torch.equal(mom, ref))
            out["sharded_equals_single_process"] = ok
        okt = torch.tensor([1 if ok else 0], device="cuda")
        dist.broadcast(okt, 0)
        assert int(okt.item()) == 1, "cfg 4: window-sharded latents differ from the single-process result"

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
    if rank == 0 and not args.no_roofline and not args.dtype.startswith("f32"):
        vae.enable_hip_graphs(False)  # per-launch timing needs the eager launches
        agg = roofline_pass(step)
        name, (fl, sec, n, fx, fl_main) = max(agg.items(), key=lambda kv: kv[1][1])
        ach = fl / sec / 1e12
        # HBM traffic per launch of that kernel: from the committed PMC passes of this same command (profiles/)
        traffic = None
        short = name.split("_", 3)[3] if name.count("_") >= 3 else name
        profiled = args.workload.startswith("cfg3") and args.dtype == "bf16"  # the committed PMC passes are of THIS command
        tj = profile_json("pmc_traffic.json") if profiled else None
        from cvvae_amd import _lib as _L
        fp_now = _L.source_fingerprint()

        def stale(doc):  # the committed counters were measured on other kernel sources than the ones running now
            return doc.get("library_source_fingerprint") != fp_now
        if tj:
            k = tj.get("kernels", {}).get(short)
            if k:
                traffic = k["fetch_bytes"] + k["write_bytes"]
            if "step_total_bytes" in tj:
                out["hbm"]["pmc_bytes_per_step"] = tj["step_total_bytes"]
                out["hbm"]["pmc_gbps"] = round(tj["step_total_bytes"] / (elapsed / args.steps) / 1e9, 1)
        out["roofline"] = {
            "bound": "mfma", "kernel": name, "launches_per_step": n,
            "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
            "executed": round(fx / sec / 1e12, 1),
            "avg_launch_ms": round(sec / n * 1e3, 4), "alg_gflop_per_launch": round(fl / n / 1e9, 2), "traffic": traffic,
            "traffic_unit": "bytes/launch (PMC, profiles/pmc_traffic.json)",
            "traffic_source": None if not tj else {"file": "profiles/pmc_traffic.json", "measured_in_this_run": False,
                                                   "library_source_fingerprint": tj.get("library_source_fingerprint"),
                                                   "running_source_fingerprint": fp_now, "stale": stale(tj)},
        }
        if fl_main < fl:  # launches of this instance are split in two dispatches (odd frame count under a two-frame tile)
            out["roofline"]["dispatches_per_launch"] = 2
            out["roofline"]["main_dispatch_alg_gflop"] = round(fl_main / n / 1e9, 2)
            out["roofline"]["launch_note"] = (
                "avg_launch_ms spans BOTH dispatches of a cvvae_conv_fwd call (HIP events around the call): the two-frame-tile kernel "
                "over all but the last frame + its one-frame-tile twin on the last frame.  A rocprofv3 trace lists them as two kernels "
                "whose average durations add up to it; against the main kernel's rocprofv3 duration use main_dispatch_alg_gflop")
        pk = profile_json("peak_probe.json")
        if pk:  # what dense bf16 matrix code sustains on this pool's MI355X (vendor GEMM, register-only MFMA stream)
            out["roofline"]["on_box_denominators"] = pk
            out["roofline"]["executed_frac_of_hipblaslt_gemm"] = round(fx / sec / 1e12 / max(pk["hipblaslt_bf16_gemm_tflops"]), 3)
        sq = profile_json("pmc_sq.json") if profiled else None
        if sq and short in sq.get("kernels", {}):
            out["mfma_busy"] = dict(sq["kernels"][short], source="profiles/pmc_sq.json (SQ counters, separate --pmc passes)",
                                    measured_in_this_run=False, stale=stale(sq))
        tot_fx, tot_sec = sum(v[3] for v in agg.values()), sum(v[1] for v in agg.values())
        out["executed_tflops_conv_kernels"] = round(tot_fx / tot_sec / 1e12, 1)
        out["kernels"] = {k: {"tflops": round(v[0] / v[1] / 1e12, 1), "executed_tflops": round(v[3] / v[1] / 1e12, 1),
                              "ms": round(v[1] * 1e3, 3), "launches": v[2]}
                          for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    if rank == 0 and not args.no_parity and args.workload in GOLDEN_OF and \
            os.path.isfile(os.path.join(P.GOLDEN_DIR, GOLDEN_OF[args.workload] + ".npz")):
        vae.enable_hip_graphs(False)
        r = P.measure(vae, GOLDEN_OF[args.workload])
        out["parity"] = {
            "against": f"tests/golden/{GOLDEN_OF[args.workload]}.npz = the reference's own modules, CPU fp32, same seeded weights/input",
            "latent_max_abs": float(f"{r['latent_max_abs']:.3e}"), "latent_mean_abs": float(f"{r['latent_mean_abs']:.3e}"),
            "recon_psnr_db": round(r["recon_psnr_db"], 2), "recon_max_abs": float(f"{r['recon_max_abs']:.3e}"),
            "reference_own_noise_same_dtype": reference_noise(GOLDEN_OF[args.workload], args.dtype),
            "north_star_tolerance": "|delta| <= 1e-3 on latents",
            "meets_north_star_tolerance": bool(r["latent_max_abs"] <= 1e-3),
        }
    if rank == 0 and world == 1 and not args.no_tolerance_mode and not args.no_parity and not args.dtype.startswith("f32") and \
            not cfg5 and args.workload in GOLDEN_OF and os.path.isfile(os.path.join(P.GOLDEN_DIR, GOLDEN_OF[args.workload] + ".npz")):
        # the SAME workload on the cheapest mode that meets the latent bound as a maximum: throughput and tolerance of one mode,
        # measured in this run next to the bench dtype (the precision ladder)
        del vae
        torch.cuda.empty_cache()
        vq = cls()
        P.load_seeded(vq, 0)
        vq = vq.float().cuda().eval()
        vq.fp32_mode = "fast"
        xq = x.float()
        for _ in range(2):
            vq.decode(vq.encode(xq).latent_dist.mode())
        torch.cuda.synchronize()
        tq = time.perf_counter()
        nq = max(2, min(args.steps, 5))
        for _ in range(nq):
            vq.decode(vq.encode(xq).latent_dist.mode())
        torch.cuda.synchronize()
        tq = (time.perf_counter() - tq) / nq
        rq = P.measure(vq, GOLDEN_OF[args.workload])
        out["tolerance_mode"] = {
            "dtype": "f32q", "what": "fp32 model, every product = fp16 MFMA + bf8 / fp6 correction MFMA (fp32_mode='fast'); bench.py --dtype f32q",
            "value": round(B * T / tq, 3), "unit": "frames/s", "ms_per_step": round(tq * 1e3, 3),
            "latent_max_abs": float(f"{rq['latent_max_abs']:.3e}"), "latent_mean_abs": float(f"{rq['latent_mean_abs']:.3e}"),
            "recon_psnr_db": round(rq["recon_psnr_db"], 2), "meets_north_star_tolerance": bool(rq["latent_max_abs"] <= 1e-3),
            "ratio_to_bench_dtype": round((B * T / tq) / out["value"], 3),
        }
        del vq
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(family, T, H, W, full=args.cpu_baseline_full)
    if rank == 0 and world == 1:
        out["reference_ops_on_this_gpu_model"] = torch_rocm_baseline(family, [B, 3, T, H, W], args.dtype, out["ms_per_step"])
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
