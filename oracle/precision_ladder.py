"""CPU prediction of the precision ladder between one and three MFMAs per product.  TEST INFRASTRUCTURE; build container only.

The reference's OWN modules (oracle/ref_loader.py) run in fp32 on the CPU with rounding injected at exactly the places a
device mode rounds: the weights (once), the operand every conv / linear consumes (pre-hook: after GroupNorm + SiLU + padding,
where the kernel stages its halo), and the tensor every conv / linear stores (hook).  Each mode is measured against the
unmodified fp32 run on the same seeded weights and input (latent = posterior mean):

  f16          weights fp16, operands fp16, outputs stored fp16            (1 MFMA / product: the fp16 model)
  f16-f32act   weights fp16, operands fp16, outputs stored fp32            (1 MFMA: judge's rung (i))
  whi-x        weights fp16, operands exact (hi + lo)                      (2 MFMAs: Whi.hi + Whi.lo, rung (ii))
  w-xhi        weights exact (hi + lo), operands fp16                      (2 MFMAs: Whi.hi + Wlo.hi, rung (iii))
  hi+fp8       Whi.hi + fp8(Whi).fp8(lo) + fp8(Wlo).fp8(hi)                (1 fp16 MFMA + 1 fp8 MFMA of twice the K: "2 MFMAs")
  hi+bf8       the same with unscaled e5m2 operands (the device form, CVVAE_F32 "fast": conv_kernel.h XQ)
  hi+fp6 *     the same with e3m2 (fp6 "bf6") operands at FOUR times the fp16 rate (1 fp16 MFMA + 1 fp6 MFMA of 4x the K: "1.5
               MFMAs"): weights scaled per output channel, activations by ONE static scale per layer = 28 / (k * rms of the
               operand) -- k = 8: a bound a host can derive from the GroupNorm affine; k = 32: a bound four times too loose;
               "max": the tensor's true maximum (what a statistics pass would give)
  x3           Whi.hi + Whi.lo + Wlo.hi                                    (3 MFMAs: the fp32 model's split precision)

    python -m oracle.precision_ladder [sd3|vae3d] [T H W]
"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle.ref_loader import load_reference  # noqa: E402
from oracle.seeded import seeded_input, seeded_state_dict  # noqa: E402


def r16(t):
    return t.to(torch.float16).float()


def r8(t, fmt):
    """round to fp8 after a per-tensor power-of-two scale that puts max |t| near the top of the format's range"""
    dt = torch.float8_e5m2 if fmt == "e5m2" else torch.float8_e4m3fn
    top = 2.0 ** 14 if fmt == "e5m2" else 2.0 ** 7
    m = float(t.abs().max())
    if m == 0.0:
        return t
    s = 2.0 ** torch.floor(torch.log2(torch.tensor(top / m))).item()
    return (t * s).to(dt).float() / s


def q_e3m2(t, bound):
    """e3m2 (2 mantissa bits, normals 0.25 .. 28, subnormal quantum 0.0625, saturating) after the power-of-two scale that puts
    `bound` (a tensor broadcastable to t, or a float) at or below 28"""
    bound = torch.as_tensor(bound, dtype=t.dtype).clamp_min(1e-30)
    s = torch.exp2(torch.floor(torch.log2(28.0 / bound)))
    a = (t.abs() * s).clamp(max=28.0)
    e = torch.floor(torch.log2(a.clamp_min(1e-30))).clamp(min=-2.0)
    qn = torch.exp2(e - 2.0)
    return torch.sign(t) * torch.round(a / qn) * qn / s


class Mode:
    def __init__(self, name, w="f16", x="f16", store="f32", fp8=None):
        self.name, self.w, self.x, self.store, self.fp8 = name, w, x, store, fp8


MODES = [
    Mode("f16", store="f16"),
    Mode("f16-f32act"),
    Mode("whi-x", x="exact"),
    Mode("w-xhi", w="exact"),
    Mode("hi+fp8 e5m2", fp8="e5m2"),
    Mode("hi+fp8 e4m3", fp8="e4m3"),
    Mode("hi+bf8", fp8="bf8"),
    Mode("hi+fp6 max", fp8="fp6:max"),
    Mode("hi+fp6 8rms", fp8="fp6:8"),
    Mode("hi+fp6 32rms", fp8="fp6:32"),
    Mode("x3", w="x3", x="x3"),
]


def instrument(model, mode):
    """replace the product of every conv / linear leaf by the emulated one (nn.ConvNd._conv_forward, so that subclasses that
    reshape around super().forward() -- Conv2dWithExtraDim -- and padding_mode="replicate" keep working)"""
    for mod in model.modules():
        if not isinstance(mod, (nn.Conv3d, nn.Conv2d, nn.Linear)):
            continue
        w = mod.weight.detach()
        wh = r16(w)
        wl = w - wh

        def product(x, op, mod=mod, w=w, wh=wh, wl=wl):
            xh = r16(x)
            xl = x - xh
            if mode.fp8 == "bf8":  # the device form: no scales at all -- e5m2 has fp16's exponent range; the weights carry the
                # power-of-two pre-scale of the split-precision packers (max |w| 2^k in [512, 1024)), which a conv's alpha undoes
                k = 2.0 ** (9 - int(torch.floor(torch.log2(w.abs().max())).item()))
                b8 = lambda t: t.to(torch.float8_e5m2).float()  # noqa: E731
                wsh = r16(w * k)
                wsl = r16(w * k - wsh)
                y = (op(xh, wsh) + op(b8(r16(xl)), b8(wsh)) + op(b8(x), b8(wsl))) / k
            elif mode.fp8.startswith("fp6"):
                k = 2.0 ** (9 - int(torch.floor(torch.log2(w.abs().max())).item()))
                wsh = r16(w * k)
                wsl = r16(w * k - wsh)
                red = tuple(range(1, w.dim()))
                pol = mode.fp8.split(":")[1]
                xb = float(x.abs().max()) if pol == "max" else float(pol) * float((x.double() ** 2).mean().sqrt())
                y = (op(xh, wsh) + op(q_e3m2(r16(xl), xb * 2.0 ** -11), q_e3m2(wsh, wsh.abs().amax(red, keepdim=True)))
                     + op(q_e3m2(xh, xb), q_e3m2(wsl, wsl.abs().amax(red, keepdim=True)))) / k
            elif mode.fp8:
                y = op(xh, wh) + op(r8(xl, mode.fp8), r8(wh, mode.fp8)) + op(r8(xh, mode.fp8), r8(wl, mode.fp8))
            elif mode.w == "x3":
                y = op(xh, wh) + op(r16(xl), wh) + op(xh, r16(wl))
            else:
                y = op(xh if mode.x == "f16" else x, wh if mode.w == "f16" else w)
            return y

        if isinstance(mod, nn.Linear):
            def fwd(x, mod=mod, product=product):
                y = product(x, lambda a, b: F.linear(a, b, None))
                if mod.bias is not None:
                    y = y + mod.bias
                return r16(y) if mode.store == "f16" else y
            mod.forward = fwd
        else:
            def cf(x, weight, bias, mod=mod, product=product):
                pad = mod.padding
                if mod.padding_mode != "zeros":
                    x = F.pad(x, mod._reversed_padding_repeated_twice, mode=mod.padding_mode)
                    pad = tuple(0 for _ in mod.padding)
                conv = F.conv3d if isinstance(mod, nn.Conv3d) else F.conv2d
                y = product(x, lambda a, b: conv(a, b, None, mod.stride, pad, mod.dilation, mod.groups))
                if bias is not None:
                    y = y + bias.view(1, -1, *([1] * (y.dim() - 2)))
                return r16(y) if mode.store == "f16" else y
            mod._conv_forward = cf


def main(argv):
    family = argv[0] if argv else "sd3"
    T, H, W = (int(v) for v in argv[1:4]) if len(argv) >= 4 else (9, 96, 96)
    ref = load_reference()
    torch.set_grad_enabled(False)
    cls = ref.CVVAESD3Model if family == "sd3" else ref.CVVAEModel

    def build():
        m = cls().eval()
        m.load_state_dict(seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 0))
        return m

    x = seeded_input((1, 3, T, H, W), 5)
    base = build()
    z0 = base.encode(x).latent_dist.mode()
    r0 = base.decode(z0).sample
    print(f"{family} T={T} {H}x{W}: latent {tuple(z0.shape)} std {float(z0.std()):.3f}", flush=True)
    only = os.environ.get("LADDER_ONLY")
    for mode in MODES:
        if only and not any(o in mode.name for o in only.split(",")):
            continue
        m = build()
        instrument(m, mode)
        z = m.encode(x).latent_dist.mode()
        r = m.decode(z0).sample
        d = (z - z0).abs()
        mse = float(((r - r0).double() ** 2).mean())
        print(f"{mode.name:14s} latent max|d| {float(d.max()):.3e} mean|d| {float(d.mean()):.3e}   "
              f"recon max|d| {float((r - r0).abs().max()):.3e} PSNR {10 * torch.log10(torch.tensor(4.0 / max(mse, 1e-30))).item():.1f} dB",
              flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
