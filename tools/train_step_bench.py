"""Timing of the training-side path (cvvae_amd/grad3d.py): one step of the reference's autoencoder training
(lvdm/models/autoencoder.py:1057-1090) on the vae3d_sd3 networks -- z = encoder(x), xrec = decoder(z), a reconstruction loss,
backward into every parameter of both networks -- and the weight-gradient kernel alone on the layer shapes of such a step.
  python tools/train_step_bench.py [--T 17 --H 256 --W 256 --dtype bf16]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ev():
    return torch.cuda.Event(enable_timing=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=17)
    ap.add_argument("--H", type=int, default=256)
    ap.add_argument("--W", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--autocast", default=None, choices=["bf16", "f16"],
                    help="with --dtype f32: run the step under torch.autocast (fp32 master weights, 16-bit launches) and nudge the "
                         "masters between steps as an optimizer would (so every step re-casts and re-packs the weights)")
    ap.add_argument("--no-golden", action="store_true", help="skip the gradient comparison with the reference-generated fixture")
    ap.add_argument("--wgrad-only", action="store_true", help="time the weight-gradient layers (10 launches each) and stop")
    ap.add_argument("--cprofile", default=None, help="write a cProfile listing of one more step (host side) to this file")
    args = ap.parse_args()
    import cvvae_amd
    from cvvae_amd import ops
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    out = {"dtype": args.dtype, "clip": [1, 3, args.T, args.H, args.W]}
    # ---- the weight-gradient kernel alone (bf16 / fp16 operands; fp32 = three bf16 MFMAs per product)
    layers = [("128->128 3x3x3 causal", 128, 128, (3, 3, 3), (1, 1, 1), ((2, 0), (1, 1), (1, 1)), (args.T, args.H, args.W)),
              ("128->128 1x3x3", 128, 128, (1, 3, 3), (1, 1, 1), ((0, 0), (1, 1), (1, 1)), (args.T, args.H, args.W)),
              ("256->256 3x3x3 causal", 256, 256, (3, 3, 3), (1, 1, 1), ((2, 0), (1, 1), (1, 1)), ((args.T + 1) // 2, args.H // 2, args.W // 2)),
              ("512->512 3x3x3 causal", 512, 512, (3, 3, 3), (1, 1, 1), ((2, 0), (1, 1), (1, 1)), ((args.T + 1) // 2, args.H // 4, args.W // 4)),
              ("128->128 3x3x3 stride 2", 128, 128, (3, 3, 3), (2, 2, 2), ((2, 0), (1, 1), (1, 1)), (args.T, args.H, args.W))]
    out["wgrad"] = []
    for name, ci, co, k, st, pad, (T, H, W) in layers:
        a = torch.randn(1, T, H, W, ci, device="cuda").to(dtype)
        To = (T + pad[0][0] + pad[0][1] - k[0]) // st[0] + 1
        Ho = (H + 2 - k[1]) // st[1] + 1
        Wo = (W + 2 - k[2]) // st[2] + 1
        g = torch.randn(1, To, Ho, Wo, co, device="cuda").to(dtype)
        kw = dict(stride=st, pad=pad, pad_mode_t=1, pad_mode_hw=1 if k[0] == 3 else 0)
        ops.conv_wgrad(a, g, k, **kw)
        e0, e1 = ev(), ev()
        e0.record()
        reps = 10 if args.wgrad_only else 3
        for _ in range(reps):
            ops.conv_wgrad(a, g, k, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * To * Ho * Wo * co * ci * k[0] * k[1] * k[2]
        out["wgrad"].append({"layer": name, "in": [T, H, W], "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 1),
                             "frac_of_mfma_peak": round(fl / ms / 1e9 / 2500.0, 4)})
        del a, g
    if args.wgrad_only:
        print(json.dumps(out))
        return
    # ---- one training step of both networks
    torch.manual_seed(0)
    m = cvvae_amd.CVVAESD3Model()  # (the classes initialise their parameters with PyTorch's default-init statistics)
    m = m.to(dtype).cuda().train()
    x = (torch.rand((1, 3, args.T, args.H, args.W)) * 2 - 1).to(dtype).cuda()
    zc = m.encoder.conv_out.weight.shape[0] // 2

    def step():
        t = [ev() for _ in range(4)]
        m.zero_grad(set_to_none=True)
        host.append(time.perf_counter())
        t[0].record()
        with torch.autocast("cuda", dtype=acd or torch.bfloat16, enabled=acd is not None):
            mom = m.encoder(x)
            z = mom[:, :zc].contiguous()
            t[1].record()
            xrec = m.decoder(z)
        t[2].record()
        loss = (xrec.float() - x.float()).pow(2).mean()
        loss.backward()
        t[3].record()
        host.append(time.perf_counter())
        torch.cuda.synchronize()
        host.append(time.perf_counter())
        if acd is not None:
            with torch.no_grad():  # (an SGD-sized nudge of the masters: the next step sees changed weights)
                for p in m.parameters():
                    p.add_(p.grad, alpha=-1e-6)
            torch.cuda.synchronize()
        return [t[i].elapsed_time(t[i + 1]) for i in range(3)], float(loss)
    host = []
    acd = {"bf16": torch.bfloat16, "f16": torch.float16, None: None}[args.autocast]
    out["autocast"] = args.autocast
    step()
    ts = [step() for _ in range(args.steps)]
    enc_f = sum(t[0][0] for t in ts) / len(ts)
    dec_f = sum(t[0][1] for t in ts) / len(ts)
    bwd = sum(t[0][2] for t in ts) / len(ts)
    h = host[3:]  # (step 0 = warm-up): per step [start, everything launched, GPU drained]
    host_launch = sum(h[i + 1] - h[i] for i in range(0, len(h), 3)) / len(ts) * 1e3
    host_wait = sum(h[i + 2] - h[i + 1] for i in range(0, len(h), 3)) / len(ts) * 1e3
    if args.cprofile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        step()
        pr.disable()
        with open(args.cprofile, "w") as f:
            pstats.Stats(pr, stream=f).sort_stats("tottime").print_stats(45)
            pstats.Stats(pr, stream=f).sort_stats("cumtime").print_stats(45)
    with torch.no_grad(), torch.autocast("cuda", dtype=acd or torch.bfloat16, enabled=acd is not None):
        m.eval()
        e0, e1 = ev(), ev()
        m.decoder(m.encoder(x)[:, :zc].contiguous())
        e0.record()
        m.decoder(m.encoder(x)[:, :zc].contiguous())
        e1.record()
        torch.cuda.synchronize()
        inf = e0.elapsed_time(e1)
    ngrad = sum(1 for p in m.parameters() if p.grad is not None)
    out["train_step"] = {"encoder_forward_taped_ms": round(enc_f, 2), "decoder_forward_taped_ms": round(dec_f, 2),
                         "backward_ms": round(bwd, 2), "inference_forward_ms": round(inf, 2),
                         "backward_over_forward": round(bwd / (enc_f + dec_f), 2), "loss": ts[-1][1],
                         "host_launch_ms": round(host_launch, 2), "host_wait_for_gpu_ms": round(host_wait, 2),
                         "parameters_with_grad": ngrad, "parameters": sum(1 for _ in m.parameters()),
                         "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}
    # ---- the gradients of a step of THIS size against the reference's own modules (tests/golden/grad_sd3_t17_256.npz:
    #      oracle/make_golden.py grad): the checker leg, after every timed region
    gpath = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "grad_sd3_t17_256.npz")
    if (args.T, args.H, args.W) == (17, 256, 256) and acd is None and os.path.isfile(gpath) and not args.no_golden:
        from oracle import parity as P
        del m
        torch.cuda.empty_cache()
        mg = cvvae_amd.CVVAESD3Model()
        P.load_seeded(mg, 0)
        mg = mg.to(dtype).cuda().train()
        out["gradient_parity_vs_reference_modules"] = dict(
            P.measure_backward(mg), fixture="tests/golden/grad_sd3_t17_256.npz (the reference's Encoder3D / Decoder3D under torch.autograd, "
                                            "CPU fp32, seeded weights, seeded cotangent)", metric="relative L2")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
