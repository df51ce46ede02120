#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: encode+decode frames/sec of the CV-VAE codec at T=17, 512x512.

One "step" = one pass of the hot path over one clip: `vae.encode(x).latent_dist.mode()` then `vae.decode(z).sample`
for a synthetic [1,3,17,512,512] clip (config.workload = BASELINE cfg 3: vae3d_sd3, bf16).  Inputs are resident in
HBM before the timed region.

N > 1: one process per GPU over RCCL.  `python bench.py --gpus N` launches the N ranks itself (it re-executes under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`); started by an external launcher
(RANK / WORLD_SIZE in the environment) it simply is one of the ranks.  The DEFAULT workload at N > 1 is north_star's temporal
shard: ONE clip of 1 + 16 N frames at 512x512 whose frames arrive SHARDED ON TIME (rank r holds only its own 16 -- rank 0: 17 --
frames), the causal halo frame exchanged point-to-point over RCCL (`batch_isend_irecv`), every rank coding its own 17-frame
window, the latents all-gathered, the pixels left sharded (cvvae_amd/dist.py `codec_step_time_sharded`; weak scaling: the
per-GPU work is cfg 3's window whatever N is, and at N = 1 the step IS cfg 3).  After the timed region every rank checks its
slice bit for bit against the single-process wrapper run on the whole clip.  Annex `temporal_shard_cfg4` (N > 1): BASELINE cfg 4
(T = 129, 720x1280) time-sharded the same way -- strong scaling, with the single-process time measured in the same invocation.
`--workload cfg3_sd3_T17_512` (or cfg2 / cfg5) at N > 1 keeps the older independent-clips mode (no data-path collective).
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
    # the default at N > 1: ONE clip of 1 + 16 N frames, time-sharded input, halo frame over RCCL send/recv (T is set from N)
    "temporal_shard_sd3_512": ("sd3", 1, None, 512, 512),
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
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: cfg3_sd3_T17_512 on one GPU, temporal_shard_sd3_512 (the same window per GPU, ONE clip "
                         "sharded on time, halo exchange over RCCL) on N > 1")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32", "f32q"],
                    help="model dtype; f32 = fp32 model in split precision (three fp16 MFMAs per product), f32q = fp32 model in fast "
                         "split precision (fp16 MFMA + bf8 / fp6 correction MFMA): the cheapest mode inside north_star's 1e-3 latent bound")
    ap.add_argument("--no-tolerance-mode", action="store_true",
                    help="skip the annex that times and checks the f32q model (the mode that meets |delta| <= 1e-3) beside the bench dtype")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-sample", action="store_true",
                    help="time the CPU path on a bounded 17x192x192 sample (about 10 s, scaled by pixel count) instead of the default: "
                         "the FULL workload shape on every host core, in a child process beside the untimed checker legs (minutes)")
    ap.add_argument("--hip-graphs", action="store_true",
                    help="replay each encoder/decoder pass as a captured hipGraph (vae.enable_hip_graphs(); for launch-bound inputs)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--verbose", action="store_true",
                    help="print the full record (per-kernel tables, counter objects, notes: ~12 KB); the default line keeps what the "
                         "contract asks for plus the encode split and the tolerance modes inside `roofline`, under 6 KB")
    ap.add_argument("--full-json", default=None, metavar="PATH",
                    help="also write the full (verbose) record to this file (default: gpurun_out/bench_full.json when gpurun_out/ exists)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="N > 1: skip the sharded == single-process bit-equality check")
    ap.add_argument("--no-cfg4-annex", action="store_true", help="N > 1: skip the temporal_shard_cfg4 strong-scaling annex")
    ap.add_argument("--cpu-baseline-wall", type=float, default=420.0,
                    help="wall-time guard (s) of the in-run full-size CPU baseline; beyond it the kept measurement is reported")
    ap.add_argument("--cpu-baseline-worker", nargs=4, metavar=("FAMILY", "T", "H", "W"), default=None, help=argparse.SUPPRESS)
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
    # fp32 models: MFMA time UNITS per time group of `sp` taps (unit = one 32x32x16 fp16 MFMA): exact = three fp16 MFMAs per tap;
    # fast = one fp16 MFMA per tap + one K = 64 correction MFMA per PAIR of taps, which lasts two units in bf8 and one in fp6
    from cvvae_amd import _lib as L_
    if d.dtype == L_.F32:
        sp = 3 * sp
    elif d.dtype == L_.F32Q:
        sp = sp + 2 * ((sp + 1) // 2)
    elif d.dtype == L_.F32Q6:
        sp = sp + (sp + 1) // 2
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


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            return next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        return ""


def cpu_baseline_measure(family, T=17, H=512, W=512, full=True):
    """The CPU path timed on this host's cores.  full=True: the workload's OWN shape on every host core -- no extrapolation (minutes
    of CPU time; bench.py runs it in a child process beside its untimed checker legs).  full=False: a bounded sample (the same
    network on a 17-frame window at 192x192, scaled by pixel count; it flatters the CPU ~3x: oneDNN runs small frames more
    efficiently).  With /root/reference mounted (the build container) the reference's OWN modules are timed (kind "reference");
    on the GPU box, where it does not exist, the oracle restatement of them (kind "port")."""
    import torch
    from oracle.seeded import seeded_input, seeded_state_dict
    from oracle.shapes import state_dict_shapes

    # oneDNN conv at the sample size stops scaling (and regresses) beyond ~32 threads; the full shape takes every core
    cores = (os.cpu_count() or 1) if full else min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    hh, ww = (H, W) if full else (min(H, 192), min(W, 192))
    x = seeded_input((1, 3, T, hh, ww), 0)
    kind, model = "port", None
    if os.path.isdir("/root/reference/models"):
        try:
            from oracle.ref_loader import load_reference
            ref = load_reference()
            model = (ref.CVVAESD3Model if family == "sd3" else ref.CVVAEModel)().eval()
            model.load_state_dict(seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, 0), strict=True)
            kind = "reference"
        except Exception:  # noqa: BLE001  (the port is the fallback wherever the reference cannot be imported)
            model = None
    with torch.no_grad():
        if model is not None:
            t0 = time.time()
            post = model.encode(x).latent_dist
            t1 = time.time()
            rec = model.decode(post.mode()).sample
            dt = time.time() - t0
        else:
            from oracle import cvvae_oracle as O
            sd = seeded_state_dict(state_dict_shapes(family), 0)
            t0 = time.time()
            mom = O.encode_moments(x, sd, {}, family)
            t1 = time.time()
            rec = O.decode_sample(O.posterior_mode(mom), sd, {}, family)
            dt = time.time() - t0
    assert rec.shape == x.shape
    area_scale = (H * W) / float(hh * ww)
    who = "the reference's own modules (PyTorch-CPU fp32)" if kind == "reference" else "oracle (PyTorch-CPU fp32 restatement)"
    return {
        "value": round(T / (dt * area_scale), 5), "unit": "frames/s", "cores": cores, "kind": kind,
        "extrapolated": area_scale != 1.0, "cpu_model": _cpu_model(),
        "encode_s": round(t1 - t0, 2), "decode_s": round(dt - (t1 - t0), 2),
        "sample": (f"{who} encode+decode of the full 1x3x{T}x{hh}x{ww} workload in {dt:.1f}s" if area_scale == 1.0 else
                   f"{who} encode+decode of 1x3x{T}x{hh}x{ww} in {dt:.1f}s; value = {T} frames / (t * {area_scale:.1f}) "
                   f"i.e. scaled by pixel count to the {H}x{W} workload"),
    }


def cpu_baseline_start(family, T, H, W):
    """start the full-size CPU measurement in a child process (so that it runs BESIDE the untimed checker legs of this run, never
    beside a timed region); returns the Popen"""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", family, str(T), str(H), str(W)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")  # the child never touches the GPU
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True), time.time()


def cpu_baseline_collect(handle, wall, family, T, H, W):
    """the child's measurement if it arrives within `wall` seconds of its start; otherwise the KEPT full-size measurement of this
    CPU model (profiles/cpu_baseline_full.json) marked measured_in_this_run = false, or -- no kept file -- the bounded sample"""
    proc, t_start = handle
    out, why = None, None
    try:
        so, _ = proc.communicate(timeout=max(1.0, wall - (time.time() - t_start)))
        line = [ln for ln in so.splitlines() if ln.startswith("{")]
        if proc.returncode == 0 and line:
            out = json.loads(line[-1])
            out["measured_in_this_run"] = True
            out["concurrent_with"] = "this run's untimed parity / roofline legs (one host thread + the GPU); never a timed region"
        else:
            why = f"worker exit code {proc.returncode}"
    except subprocess.TimeoutExpired:
        proc.kill()
        why = f"not finished within the {wall:.0f} s wall guard (--cpu-baseline-wall)"
    if out is None:
        kept = os.path.join(ROOT, "profiles", "cpu_baseline_full.json")
        k = json.load(open(kept)) if os.path.isfile(kept) else None
        if k is not None and k.get("cpu_model") == _cpu_model():
            out = dict(k, measured_in_this_run=False, why_not=why)
        else:
            out = dict(cpu_baseline_measure(family, T, H, W, full=False), measured_in_this_run=True, why_not_full=why)
    return out


def _rccl_version(torch):
    try:  # (diagnostic only: never let it take the bench line down)
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:  # noqa: BLE001
        return f"unavailable ({type(e).__name__})"


# ------------------------------------------------------------------------------------------------------------------------
# the temporal shard (N > 1 default): module-level so that tests/test_dist_gloo.py drives EXACTLY the step bench.py times
# ------------------------------------------------------------------------------------------------------------------------
def temporal_shard_T(world: int, stride: int = 16) -> int:
    """frames of the sharded clip: one `stride`-frame window per rank + the clip's first frame (N = 1: cfg 3's 17)"""
    return 1 + stride * world


def temporal_shard_input(T_total, H, W, world, rank, dtype, device, seed=1000, stride=16, keep_full=True):
    """-> (x_full on the CPU or None, x_local on `device`): the SAME seeded clip on every rank, of which the rank keeps only
    its `owned_frames` on the device (time-sharded input); the full clip stays on the host for the post-run check"""
    import torch
    from cvvae_amd import dist as D
    g = torch.Generator().manual_seed(seed)
    x_full = (torch.rand((1, 3, T_total, H, W), generator=g) * 2 - 1).to(dtype)
    a, b = D.owned_frames(T_total, stride, world, rank)
    x_local = x_full[:, :, a:b].contiguous().to(device)
    return (x_full if keep_full else None), x_local


def temporal_shard_step(vae, x_local, T_total, group=None):
    """one timed step at N > 1: halo exchange (send/recv) -> encode own windows -> all_gather moments -> decode own windows"""
    from cvvae_amd import dist as D
    return D.codec_step_time_sharded(vae, x_local, T_total, group=group)


def temporal_shard_check(vae, x_full, mom, y_local, world, rank, device):
    """every rank: the single-process wrapper on the WHOLE clip must reproduce the gathered moments and this rank's frames
    bit for bit (windows are independent network calls; the halo frame is the neighbour's own data)"""
    import torch
    from cvvae_amd import dist as D
    x = x_full.to(device)
    ref_mom = vae.encode(x).latent_dist.parameters
    zc = ref_mom.shape[1] // 2
    ok = bool(torch.equal(mom, ref_mom))
    a, b = D.decoded_frames_of_rank(vae, ref_mom.shape[2], world, rank)
    if b > a:
        ref_y = vae.decode(ref_mom[:, :zc]).sample[:, :, a:b]
        ok = ok and y_local is not None and tuple(y_local.shape) == tuple(ref_y.shape) and bool(torch.equal(y_local, ref_y))
    else:
        ok = ok and y_local is None
    return ok


def init_rccl(torch, world, rank, local_rank, seconds=180):
    """process group on RCCL with a bounded first contact: init + one all_reduce under `seconds`, and a ONE-LINE diagnosis on
    stderr if it does not come back (the RCCL path of this project first ran on the driver's multi-GPU tier)."""
    import datetime
    import threading

    import torch.distributed as dist

    what = {"at": "init_process_group"}

    def watchdog():
        print(f"bench.py: rank {rank}/{world} (cuda:{local_rank}, {torch.cuda.device_count()} devices visible) still inside "
              f"{what['at']} after {seconds} s -- RCCL first contact hangs: check HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC), "
              f"MASTER_ADDR=127.0.0.1, one rank per GPU, NCCL_DEBUG=INFO for the transport it picked", file=sys.stderr, flush=True)

    tmr = threading.Timer(seconds, watchdog)
    tmr.daemon = True
    tmr.start()
    try:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank),
                                timeout=datetime.timedelta(seconds=max(seconds * 2, 600)))
        what["at"] = "the first all_reduce"
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        if int(t.item()) != world:
            raise RuntimeError(f"first all_reduce over {world} ranks returned {t.item()}")
    except Exception as e:  # noqa: BLE001
        print(f"bench.py: rank {rank}/{world}: RCCL first contact failed in {what['at']}: {type(e).__name__}: {e}", file=sys.stderr,
              flush=True)
        raise
    finally:
        tmr.cancel()
    return dist


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


def roofline_report(agg, args, sec_per_step, profiled, brief=False):
    """-> {"roofline": ..., "mfma_busy": ..., "kernels": ...} from a roofline_pass() aggregate.  `achieved` = ALGORITHMIC FLOPs
    (2 M N K of the reference op) / HIP-event time of the dominant conv instance; `executed` = the MFMA work really issued, in
    units of the fp16 32x32x16 MFMA (16-bit models: one per product and executed tap; fp32 models: three -- exact -- or 1 + 1 (bf8)
    / 1 + 1/2 (fp6) per product -- fast: conv_flops_executed).  Counter-derived fields come from the committed PMC passes of the
    cfg 3 bf16 command (`profiled`), stamped with the fingerprint of the kernel sources they were measured on."""
    out = {}
    name, (fl, sec, n, fx, fl_main) = max(agg.items(), key=lambda kv: kv[1][1])
    ach = fl / sec / 1e12
    traffic = None
    short = name.split("_", 3)[3] if name.count("_") >= 3 else name
    tj = profile_json("pmc_traffic.json") if profiled else None
    from cvvae_amd import _lib as _L
    fp_now = _L.source_fingerprint()

    def stale(doc):  # the committed counters were measured on other kernel sources than the ones running now
        return doc.get("library_source_fingerprint") != fp_now
    hbm_extra = {}
    if tj:
        k = tj.get("kernels", {}).get(short)
        if k:
            traffic = k["fetch_bytes"] + k["write_bytes"]
        if "step_total_bytes" in tj:
            hbm_extra = {"pmc_bytes_per_step": tj["step_total_bytes"], "pmc_gbps": round(tj["step_total_bytes"] / sec_per_step / 1e9, 1)}
    out["roofline"] = {
        "bound": "mfma", "kernel": name, "launches_per_step": n,
        "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
        "executed": round(fx / sec / 1e12, 1),
        "executed_note": "MFMA time units issued x 2*32*32*16 FLOP / time (fp32 models: 3 units per product exact; fast: 2 with bf8, 1.5 "
                         "with fp6 corrections) -- the matrix-pipe occupancy as a rate; `achieved` counts the reference op's FLOPs once",
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
    if pk and not brief:  # what dense bf16 matrix code sustains on this pool's MI355X (vendor GEMM, register-only MFMA stream)
        out["roofline"]["on_box_denominators"] = pk
        out["roofline"]["executed_frac_of_hipblaslt_gemm"] = round(fx / sec / 1e12 / max(pk["hipblaslt_bf16_gemm_tflops"]), 3)
    sq = profile_json("pmc_sq.json") if profiled else None
    if sq and short in sq.get("kernels", {}):
        out["mfma_busy"] = dict(sq["kernels"][short], source="profiles/pmc_sq.json (SQ counters, separate --pmc passes)",
                                measured_in_this_run=False, stale=stale(sq))
    tot_fl, tot_fx, tot_sec = sum(v[0] for v in agg.values()), sum(v[3] for v in agg.values()), sum(v[1] for v in agg.values())
    out["executed_tflops_conv_kernels"] = round(tot_fx / tot_sec / 1e12, 1)
    out["algorithmic_tflops_conv_kernels"] = round(tot_fl / tot_sec / 1e12, 1)
    out["kernels"] = {k: {"tflops": round(v[0] / v[1] / 1e12, 1), "executed_tflops": round(v[3] / v[1] / 1e12, 1),
                          "ms": round(v[1] * 1e3, 3), "launches": v[2]}
                      for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    if hbm_extra:
        out["hbm_pmc"] = hbm_extra
    return out


def main():
    args = parse()
    if args.cpu_baseline_worker is not None:  # child of cpu_baseline_start(): CPU only, one JSON line
        fam, T_, H_, W_ = args.cpu_baseline_worker
        print(json.dumps(cpu_baseline_measure(fam, int(T_), int(H_), int(W_), full=True)), flush=True)
        return
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
    dist = init_rccl(torch, world, rank, local_rank) if world > 1 else None

    import cvvae_amd
    from cvvae_amd import dist as D
    from oracle import parity as P  # checker only: seeded weights + the fixture comparison (never inside the timed region)

    if args.workload is None:
        args.workload = "cfg3_sd3_T17_512" if world == 1 else "temporal_shard_sd3_512"
    tshard = args.workload.startswith("temporal_shard")
    if tshard and world == 1:
        args.workload, tshard = "cfg3_sd3_T17_512", False  # one rank: the shard IS cfg 3 (T = 1 + 16)
    family, B, T, H, W = WORKLOADS[args.workload]
    if tshard:
        T = temporal_shard_T(world)
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32q": torch.float32}[args.dtype]
    cls = cvvae_amd.CVVAESD3Model if family == "sd3" else cvvae_amd.CVVAEModel
    vae = cls()
    P.load_seeded(vae, 0)  # random weights of the named architecture with PyTorch's default-init statistics (no checkpoint access)
    # frozen, as the reference's script loads it (cvvae_inference_video.py:11-12: `.cuda()`, `requires_grad_(False)`): a network with
    # trainable parameters re-checks its parameter checksum -- one host sync -- on every inference pass (modeling._Net.forward)
    vae = vae.to(dtype).cuda().eval().requires_grad_(False)
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
    x_full = None
    if tshard:
        # ONE clip, the same on every rank (same seed); the rank keeps only ITS frames on the device
        x_full, x = temporal_shard_input(T, H, W, world, rank, dtype, "cuda", keep_full=not args.no_check)
    else:
        # cfg 3 / 2: every rank codes its own clip (weak scaling).  cfg 4: every rank holds the same clip (seed without rank)
        g = torch.Generator().manual_seed(1000 + (0 if cfg4 else rank))
        x = (torch.rand((B, 3, T, H, W), generator=g) * 2 - 1).to(dtype).cuda()

    if tshard:
        def step():
            return temporal_shard_step(vae, x, T)
    elif cfg4 and dist is not None:
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
    D.TRAFFIC.update(sent=0, recv=0)
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
    assert tshard or cfg5 or (cfg4 and dist is not None) or y.shape == x.shape
    rank_times = None
    traffic = None
    if dist is not None:
        t = torch.tensor([elapsed, float(D.TRAFFIC["sent"]), float(D.TRAFFIC["recv"])], device="cuda", dtype=torch.float64)
        ts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(ts, t)                      # per-rank clocks and wire bytes (diagnostic); the reported time is their MAX
        rank_times = [round(float(v[0].item()) / args.steps * 1e3, 3) for v in ts]
        traffic = {"bytes_sent_per_step_by_rank": [int(v[1].item()) // max(args.steps, 1) for v in ts],
                   "bytes_recv_per_step_by_rank": [int(v[2].item()) // max(args.steps, 1) for v in ts]}
        elapsed = max(float(v[0].item()) for v in ts)

    strong = cfg4                                 # one fixed clip split over the ranks; everything else: per-rank work fixed
    if tshard:
        frames = T * args.steps                   # ONE clip of 1 + 16 N frames per step, over all ranks
        units = world * args.steps                # every rank codes one cfg-3 window (rank r > 0 re-codes its halo frame)
    else:
        frames = (1 if strong else world) * B * T * args.steps
        units = (1 if strong else world) * (B if cfg5 else 1) * args.steps / (8 if cfg5 else 1)  # ALG_TFLOP units done
    alg_key = "cfg3_sd3_T17_512" if tshard else args.workload
    if tshard:
        par = (f"ONE clip of {T} = 1 + 16 x {world} frames sharded on time over {world} ranks: halo frame by RCCL send/recv "
               f"(batch_isend_irecv), one 17-frame window per rank, moments all-gathered, pixels left sharded")
    elif strong:
        par = f"(temporal window x spatial tile) network calls sharded x{world}, latents all-gathered"
    else:
        par = f"independent clips x{world} (no collective)"
    out = {
        "metric": "encode+decode frames/sec (T=17, 512x512)" if (args.workload.startswith("cfg3") or tshard) else
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
        "config": {"workload": (f"{args.workload}: {family} encode(x).mode() + decode(z) of ONE [1,3,{T},{H},{W}] clip, rank r holding "
                                f"frames owned_frames({T}, 16, {world}, r) -- a 17-frame window (cfg 3's) per GPU" if tshard else
                                f"{args.workload}: {family} " + ("encode(x).mode()" if cfg5 else "encode(x).mode() + decode(z)") +
                                f", x=[{B},3,{T},{H},{W}] per GPU"),
                   "clips_per_gpu": (round(1.0 / world, 4) if tshard else B), "hip_graphs": bool(args.hip_graphs), "parallelism": par},
        "multi_gpu": None if dist is None else dict({
            "n_ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
            "rccl_version": _rccl_version(torch),
            "ms_per_step_by_rank": rank_times, "devices": torch.cuda.device_count(),
            "device_name": torch.cuda.get_device_name(local_rank),
            "data_path_collectives": (
                "per step: 1 batched send/recv of the boundary pixel frame to the right neighbour (1.5 MB in bf16 at 512x512) + 1 "
                "all_gather of the posterior moments on time (padded to the largest rank's 5 latent frames)" if tshard else
                ("all_gather of the latents (15 MB) once per step; tile results point-to-point only when a window's tiles span "
                 "ranks") if strong else "none (independent clips)"),
            "note": "value = work of all ranks / MAX over ranks of the barrier-bracketed time"}, **(traffic or {})),
        "achieved_tflops_whole_path": round(ALG_TFLOP[alg_key] * units / elapsed, 1),
        "hbm": {"algorithmic_gbps": round(ALG_GB[alg_key] * units / elapsed, 1), "peak_gbps": HBM_PEAK_GBPS,
                "note": "algorithmic bytes (BASELINE.md section 3) / step time; the path is MFMA-bound (AI ~1900 FLOP/B)"},
    }

    if dist is not None and not args.no_check and (tshard or cfg4):
        # the sharded run must reproduce the single-process wrapper bit for bit (windows are independent network calls)
        if tshard:
            mom, y_loc = step()
            ok = temporal_shard_check(vae, x_full, mom, y_loc, world, rank, "cuda")
            del mom, y_loc
        else:
            mom = D.encode_units_sharded(vae, x)
            ok = bool(torch.equal(mom, vae.encode(x).latent_dist.parameters)) if rank == 0 else True
        okt = torch.tensor([1 if ok else 0], device="cuda")
        oks = [torch.empty_like(okt) for _ in range(world)]
        dist.all_gather(oks, okt)
        out["sharded_equals_single_process"] = all(int(v.item()) == 1 for v in oks)
        out["sharded_check"] = ("every rank: gathered moments and its own decoded frames == the single-process wrapper on the whole "
                                "clip, bit for bit" if tshard else "rank 0: gathered moments == the single-process wrapper, bit for bit")
        assert out["sharded_equals_single_process"], f"sharded results differ from the single-process run: {[int(v.item()) for v in oks]}"
    x_full = None

    if dist is not None and tshard and not args.no_cfg4_annex:
        # BASELINE cfg 4 through the same time-sharded step: ONE fixed clip (T = 129, 720x1280: 8 windows x 6 blended spatial
        # tiles), strong scaling, with the single-process time of the same clip measured HERE on rank 0
        fam4, _, T4, H4, W4 = WORKLOADS["cfg4_sd3_T129_720x1280"]
        x4_full, x4 = temporal_shard_input(T4, H4, W4, world, rank, dtype, "cuda", seed=1004, keep_full=(rank == 0))
        temporal_shard_step(vae, x4, T4)             # warmup
        torch.cuda.synchronize()
        dist.barrier()
        t4 = time.perf_counter()
        n4 = 2
        for _ in range(n4):
            temporal_shard_step(vae, x4, T4)
        torch.cuda.synchronize()
        mine4 = (time.perf_counter() - t4) / n4
        dist.barrier()
        tt = torch.tensor([mine4], device="cuda", dtype=torch.float64)
        t4s = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(t4s, tt)
        t4max = max(float(v.item()) for v in t4s)
        single = None
        if rank == 0:
            xg = x4_full.cuda()
            vae.decode(vae.encode(xg).latent_dist.mode())   # warmup of the single-process shapes
            torch.cuda.synchronize()
            ts1 = time.perf_counter()
            vae.decode(vae.encode(xg).latent_dist.mode())
            torch.cuda.synchronize()
            single = time.perf_counter() - ts1
            del xg
        dist.barrier()
        out["temporal_shard_cfg4"] = {
            "workload": f"cfg4_sd3_T129_720x1280: ONE [1,3,{T4},{H4},{W4}] clip, time-sharded over {world} ranks (8 windows; 6 blended "
                        f"spatial tiles per window on the owning rank), encode + decode, moments all-gathered",
            "scaling": "strong", "steps": n4, "ms_per_step": round(t4max * 1e3, 2), "value": round(T4 / t4max, 2), "unit": "frames/s",
            "ms_per_step_by_rank": [round(float(v.item()) * 1e3, 2) for v in t4s],
            "single_process_ms_same_invocation": None if single is None else round(single * 1e3, 2),
            "speedup_vs_single_process": None if single is None else round(single / t4max, 3),
        }
        del x4, x4_full
        torch.cuda.empty_cache()

    if rank == 0 and not (tshard or cfg5 or (cfg4 and dist is not None)):
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
    roof_step = step
    if dist is not None:
        # rank 0 alone runs the per-launch timing pass: it must not enter a collective.  The temporal shard's per-GPU work is one
        # 17-frame window -- rank 0's own frames ARE such a clip; other multi-GPU workloads: no roofline pass
        if tshard:
            def roof_step():
                return vae.decode(vae.encode(x).latent_dist.mode()).sample
        elif cfg4:
            roof_step = None
    if rank == 0 and not args.no_roofline and roof_step is not None:
        vae.enable_hip_graphs(False)  # per-launch timing needs the eager launches
        out.update(roofline_report(roofline_pass(roof_step), args, elapsed / args.steps,
                                   profiled=args.workload.startswith("cfg3") and args.dtype == "bf16"))
        out["hbm"].update(out.pop("hbm_pmc", {}))
    golden = GOLDEN_OF.get("cfg3_sd3_T17_512" if tshard else args.workload)  # (the shard's per-GPU window is cfg 3's clip shape)
    have_golden = golden is not None and os.path.isfile(os.path.join(P.GOLDEN_DIR, golden + ".npz"))
    tol_mode = (rank == 0 and world == 1 and not args.no_tolerance_mode and not args.no_parity and not args.dtype.startswith("f32")
                and not cfg5 and have_golden)
    vq = None
    if tol_mode:
        # the SAME workload on the cheapest mode that meets the latent bound as a maximum: throughput, roofline and tolerance of one
        # mode, measured in this run next to the bench dtype (the precision ladder).  Timed BEFORE the CPU
        # baseline child starts.
        vq = cls()
        P.load_seeded(vq, 0)
        vq = vq.float().cuda().eval().requires_grad_(False)
        vq.fp32_mode = "fast"
        xq = x.float()

        def qstep():
            return vq.decode(vq.encode(xq).latent_dist.mode()).sample
        for _ in range(2):
            qstep()
        torch.cuda.synchronize()
        tq = time.perf_counter()
        nq = max(2, min(args.steps, 5))
        for _ in range(nq):
            qstep()
        torch.cuda.synchronize()
        tq = (time.perf_counter() - tq) / nq
        out["tolerance_mode"] = {
            "dtype": "f32q", "what": "fp32 model, every product = fp16 MFMA + bf8 / fp6 correction MFMA (fp32_mode='fast'); bench.py --dtype f32q",
            "value": round(B * T / tq, 3), "unit": "frames/s", "ms_per_step": round(tq * 1e3, 3),
            "ratio_to_bench_dtype": round((B * T / tq) / out["value"], 3),
        }
        if not args.no_roofline:
            qargs = argparse.Namespace(**dict(vars(args), dtype="f32q"))
            rq_ = roofline_report(roofline_pass(qstep), qargs, tq, profiled=False, brief=True)
            rq_.pop("hbm_pmc", None)
            out["tolerance_mode"].update(rq_)
        # ... and the MIXED tolerance mode: north_star bounds the LATENTS (the encoder's output) and asks the frames to match
        # "within fp16 tolerance" -- the same fp32-fast encoder, the decoder on fp16 copies of its weights with fp16 activations
        # (model.decoder_compute_dtype)
        vq.decoder_compute_dtype = torch.float16
        evq = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for i in range(2 + nq):
            if i == 2:
                torch.cuda.synchronize()
                tm = time.perf_counter()
            evq[0].record()
            zq = vq.encode(xq).latent_dist.mode()
            evq[1].record()
            vq.decode(zq)
            evq[2].record()
        torch.cuda.synchronize()
        tm = (time.perf_counter() - tm) / nq
        out["tolerance_mode_mixed"] = {
            "dtype": "f32q encoder + f16 decoder",
            "what": "fp32 model: encoder in fp32_mode='fast' (the latents are its output: same bound as tolerance_mode), decoder on "
                    "fp16 weight copies and fp16 activations (decoder_compute_dtype=torch.float16: frames within the fp16 model's error)",
            "value": round(B * T / tm, 3), "unit": "frames/s", "ms_per_step": round(tm * 1e3, 3),
            "encode_ms": round(evq[0].elapsed_time(evq[1]), 3), "decode_ms": round(evq[1].elapsed_time(evq[2]), 3),
            "ratio_to_bench_dtype": round((B * T / tm) / out["value"], 3),
        }
        vq.decoder_compute_dtype = None
        del xq, zq
    cpu_handle = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_baseline_sample:
        out["cpu_baseline"] = dict(cpu_baseline_measure(family, T, H, W, full=False), measured_in_this_run=True)
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        # every timed region of this run is over: the full-size CPU measurement runs beside the parity legs below
        cpu_handle = cpu_baseline_start(family, T, H, W)
    if rank == 0 and not args.no_parity and have_golden:
        vae.enable_hip_graphs(False)
        r = P.measure(vae, golden)
        out["parity"] = {
            "against": f"tests/golden/{golden}.npz = the reference's own modules, CPU fp32, same seeded weights/input",
            "weights": "seeded random weights with PyTorch default-init statistics (GroupNorm gamma = 1, beta = 0): no checkpoint access",
            "latent_max_abs": float(f"{r['latent_max_abs']:.3e}"), "latent_mean_abs": float(f"{r['latent_mean_abs']:.3e}"),
            "recon_psnr_db": round(r["recon_psnr_db"], 2), "recon_max_abs": float(f"{r['recon_max_abs']:.3e}"),
            "reference_own_noise_same_dtype": reference_noise(golden, args.dtype),
            "north_star_tolerance": "|delta| <= 1e-3 on latents",
            "meets_north_star_tolerance": bool(r["latent_max_abs"] <= 1e-3),
        }
    if rank == 0 and cfg5 and not args.no_parity and os.path.isfile(os.path.join(P.GOLDEN_DIR, "cfg5slice_sd3_b2_t33_512_enc.npz")):
        # cfg 5's own fixture: a B = 2 slice of the batch of 8, both 17-frame windows of every clip, from the reference's modules
        vae.enable_hip_graphs(False)
        r = P.measure_encode(vae, "cfg5slice_sd3_b2_t33_512_enc", latents=lambda xx: vae.encode_latents(xx, sample=False))
        out["parity"] = {
            "against": "tests/golden/cfg5slice_sd3_b2_t33_512_enc.npz = the reference's own modules, CPU fp32: a B = 2 slice "
                       "[2,3,33,512,512] of the workload's batch, through the timed entry point (encode_latents, posterior mode)",
            "latent_max_abs": float(f"{r['latent_max_abs']:.3e}"), "latent_mean_abs": float(f"{r['latent_mean_abs']:.3e}"),
            "latent_max_abs_per_batch_item": [float(f"{v:.3e}") for v in r["latent_max_abs_per_batch_item"]],
            "north_star_tolerance": "|delta| <= 1e-3 on latents", "meets_north_star_tolerance": bool(r["latent_max_abs"] <= 1e-3),
        }
    if rank == 0 and cfg4 and not args.no_parity and os.path.isfile(os.path.join(P.GOLDEN_DIR, "cfg4_sd3_t129_720x1280_enc.npz")):
        # cfg 4's own fixture: the WHOLE clip's posterior mean (8 windows x 6 blended tiles) from the reference's modules, encode side
        vae.enable_hip_graphs(False)
        r = P.measure_encode(vae, "cfg4_sd3_t129_720x1280_enc")
        out["parity"] = {
            "against": "tests/golden/cfg4_sd3_t129_720x1280_enc.npz = the reference's own modules, CPU fp32: encode of the whole "
                       "[1,3,129,720,1280] clip (posterior mean sampled at stride 2), single-process wrapper",
            "latent_max_abs": float(f"{r['latent_max_abs']:.3e}"), "latent_mean_abs": float(f"{r['latent_mean_abs']:.3e}"),
            "north_star_tolerance": "|delta| <= 1e-3 on latents", "meets_north_star_tolerance": bool(r["latent_max_abs"] <= 1e-3),
        }
    if vq is not None:
        del vae
        torch.cuda.empty_cache()
        rq = P.measure(vq, golden)
        out["tolerance_mode"].update({
            "weights": "seeded random weights (GroupNorm gamma = 1, beta = 0); the fp6 correction form is taken only where the norm's "
                       "bound 8 max|gamma| + max|beta| <= 16 (cvvae_amd/engine.py act_bound), bf8 corrections elsewhere",
            "latent_max_abs": float(f"{rq['latent_max_abs']:.3e}"), "latent_mean_abs": float(f"{rq['latent_mean_abs']:.3e}"),
            "recon_psnr_db": round(rq["recon_psnr_db"], 2), "meets_north_star_tolerance": bool(rq["latent_max_abs"] <= 1e-3)})
        vq.decoder_compute_dtype = torch.float16
        rm = P.measure(vq, golden)
        out["tolerance_mode_mixed"].update({
            "latent_max_abs": float(f"{rm['latent_max_abs']:.3e}"), "latent_mean_abs": float(f"{rm['latent_mean_abs']:.3e}"),
            "recon_psnr_db": round(rm["recon_psnr_db"], 2), "meets_north_star_tolerance": bool(rm["latent_max_abs"] <= 1e-3),
            "reference_own_fp16_recon_psnr_db": (reference_noise(golden, "f16") or {}).get("recon_psnr_db")})
        del vq
    if cpu_handle is not None:
        out["cpu_baseline"] = cpu_baseline_collect(cpu_handle, args.cpu_baseline_wall, family, T, H, W)
    if rank == 0 and world == 1:
        out["reference_ops_on_this_gpu_model"] = torch_rocm_baseline(family, [B, 3, T, H, W], args.dtype, out["ms_per_step"])
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        full = args.full_json or (os.path.join(ROOT, "gpurun_out", "bench_full.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
        if full:
            try:
                with open(full, "w") as f:
                    json.dump(out, f, indent=1)
            except OSError:
                pass
        print(json.dumps(out if args.verbose else compact_record(out)))


def compact_record(out: dict) -> dict:
    """The default JSON line: everything the contract names, with the figures a reader of the driver's record needs -- the encode
    split, the encode path's fraction of the MFMA peak, both tolerance modes -- INSIDE `roofline` (the driver keeps that object
    whole and a fixed key set besides), and without the per-kernel tables, counter objects and prose notes (`--verbose`,
    gpurun_out/bench_full.json)."""
    o = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype", "data", "config") if k in out}
    r = dict(out.get("roofline") or {})
    for k in ("executed_note", "launch_note", "traffic_unit", "on_box_denominators"):
        r.pop(k, None)
    ts = r.pop("traffic_source", None)
    if ts:
        r["traffic_measured_in_this_run"], r["traffic_stale"] = ts.get("measured_in_this_run"), ts.get("stale")
    for k in ("encode_ms", "decode_ms", "encode_tflops", "encode_frac_of_mfma_peak"):
        if k in out:
            r[k] = out[k]
    mb = out.get("mfma_busy")
    if mb:
        r["mfma_busy"] = {k: mb[k] for k in ("mfma_busy", "clk_ghz", "mfma_busy_x_clk_over_nominal", "wait_inst_any_share_of_wave_cycles",
                                             "measured_in_this_run", "stale") if k in mb} or None
    for key in ("tolerance_mode", "tolerance_mode_mixed"):
        t = out.get(key)
        if t:
            r[key] = {k: t[k] for k in ("dtype", "value", "ms_per_step", "ratio_to_bench_dtype", "encode_ms", "decode_ms", "latent_max_abs",
                                        "recon_psnr_db", "meets_north_star_tolerance") if k in t}
            tr = t.get("roofline")
            if tr:
                r[key].update(kernel=tr.get("kernel"), frac=tr.get("frac"), avg_launch_ms=tr.get("avg_launch_ms"))
    # the cheapest mode inside north_star's tolerance (|delta| <= 1e-3 on the latents, frames within fp16 tolerance), by name and rate
    inside = [(r[k]["value"], k) for k in ("tolerance_mode", "tolerance_mode_mixed")
              if isinstance(r.get(k), dict) and r[k].get("meets_north_star_tolerance") and r[k].get("value")]
    if inside:
        r["fastest_mode_inside_tolerance"] = {"mode": max(inside)[1], "value": max(inside)[0]}
    if r:
        o["roofline"] = r
    cb = out.get("cpu_baseline")
    if cb:
        o["cpu_baseline"] = {k: v for k, v in cb.items() if k not in ("concurrent_with",)}
    pa = out.get("parity")
    if pa:
        o["parity"] = {k: pa[k] for k in ("latent_max_abs", "latent_mean_abs", "recon_psnr_db", "meets_north_star_tolerance") if k in pa}
        own = pa.get("reference_own_noise_same_dtype") or {}
        if own:
            o["parity"]["reference_own_noise_same_dtype"] = {k: round(float(own[k]), 5) for k in ("latent_max", "latent_mean", "recon_psnr_db") if k in own}
    mg = out.get("multi_gpu")
    if mg:
        o["multi_gpu"] = {k: v for k, v in mg.items() if k not in ("data_path_collectives", "note")}
    for k in ("achieved_tflops_whole_path", "sharded_equals_single_process", "temporal_shard_cfg4"):
        if k in out:
            o[k] = out[k]
    if "hbm" in out:
        o["hbm"] = {k: v for k, v in out["hbm"].items() if k != "note"}
    ro = out.get("reference_ops_on_this_gpu_model")
    if ro and ro.get("runs"):
        o["reference_ops_on_this_gpu_speedup"] = {str(x.get("kernels")): x.get("this_run_speedup") for x in ro["runs"]}
    return o


if __name__ == "__main__":
    main()
