"""Layer programs of the two CV-VAE families, expressed as sequences of C-ABI kernel launches on NDHWC tensors.

This is the host-side mirror of the reference's L1/L2 modules: every function names
the reference forward it reproduces.  What the reference does as separate ATen ops (F.pad, F.interpolate,
GroupNorm, SiLU, add, rearrange) is folded into the conv launches -- see include/cvvae.h.
"""
import os
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L
from . import ops

REP, ZERO = L.PAD_REPLICATE, L.PAD_ZERO
P1 = ((1, 1), (1, 1), (1, 1))          # symmetric pad 1 on T,H,W
PC = ((2, 0), (1, 1), (1, 1))          # causal: T front 2
P2D = ((0, 0), (1, 1), (1, 1))         # per-frame 3x3
P0 = ((0, 0), (0, 0), (0, 0))


class WeightCache:
    """Packed (MFMA fragment order) weights and fp32 norm parameters of one nn.Module tree, rebuilt lazily whenever
    a parameter was replaced, moved (.cuda()/.to()) or modified in place (load_state_dict)."""

    def __init__(self, module: torch.nn.Module):
        self.m = module
        self._c: Dict[str, tuple] = {}
        self._names: Dict[str, tuple] = {}      # parameter name -> (owning submodule's _parameters dict, leaf name): see p()
        self._absent: set = set()               # names has() found missing (the module structure is fixed per class)
        # fp32 models: pack multi-tap conv weights in the fast layout (CVVAE_F32Q: fp16 MFMA + bf8 correction MFMA, DESIGN.md
        # section 4) instead of the three-MFMA split-precision one; set through the model's `fp32_mode`
        self.fast = False
        # ... and, behind a GroupNorm + SiLU, with e3m2 ("fp6") corrections (CVVAE_F32Q6: 1.5x instead of 2x the MFMA time of a 16-bit
        # model); CVVAE_F32_FP6=0 keeps the bf8 form everywhere (A/B aid)
        self.fast6 = os.environ.get("CVVAE_F32_FP6", "1") != "0"
        # e3m2 spans ~9 binades under ONE power-of-two scale per launch: with a bound b the codes' normal range starts at about b / 112,
        # and post-SiLU negatives never exceed 0.278 -- a bound above ~16 would put that whole branch (and the small positives) into
        # e3m2's subnormals, where the correction terms degrade towards the fp16 model's error.  The form is MEASURED at the bound of
        # default-initialised norms (8) and on spread affines up to ~14 (tests/test_gpu_fast_fp32.py); norms whose bound exceeds this
        # cap keep the bf8 form, which needs no scale.  (No real checkpoint can be measured here: DESIGN.md section 4.)
        self.fp6_bound_cap = float(os.environ.get("CVVAE_F32_FP6_CAP", "16"))
        # torch.autocast over an fp32 model (mixed-precision training / inference: fp32 master weights, 16-bit compute): p() then hands
        # out 16-bit COPIES of the weight tensors (dim >= 2: conv kernels, linear layers), refreshed in place when the master changes;
        # biases and norm affines stay the fp32 masters (the kernels take them as fp32 anyway).  Set by modeling._Net.forward.
        self.compute_dtype: Optional[torch.dtype] = None
        self._cast: Dict[str, list] = {}        # name -> [16-bit copy, key of the master it was made from]

    def _q(self) -> str:
        return "#q" if self.fast else ""

    # ---- staleness.  Every cached form is keyed on its source parameters' (data_ptr, _version, dtype, device): load_state_dict,
    #      optimizer steps, .to() / .cuda() and assignment all move the key.  A write through `.data` does NOT (`p.data.copy_(...)`
    #      leaves `p._version` alone) -- which is exactly how the reference's EMA swaps weights for validation / log_images
    #      (LitEma.copy_to / restore, /root/reference/lvdm/modules/ema.py:61-86).  Three answers, cheapest first:
    #        * invalidate(): drop everything (the caller knows it wrote through .data): `model.refresh_weights()`;
    #        * guard(): a fused device-side checksum over the masters (three multi-tensor reductions, ONE host sync), compared with the
    #          last one.  modeling._Net.forward runs it (a) on the first pass after every train() / eval() transition, and (b) on EVERY
    #          inference-branch pass (eval() or no_grad) of a network that still has a parameter with requires_grad: LitEma.copy_to /
    #          restore write exactly those parameters, and the reference's validation_step / log_images run a plain pass FIRST and enter
    #          ema_scope() afterwards, in the same mode (lvdm/models/autoencoder.py:379-384, 1193, 1426) -- a check on transitions alone
    #          would hand the EMA pass the live weights' packed forms.  A frozen model (`requires_grad_(False)`: what
    #          cvvae_inference_video.py:12 does) cannot be written by the EMA and pays no sync per pass; `weight_guard` forces (b);
    #        * the taped training pass (grad3d.run_trainable) checks on transitions only: optimizer steps move `_version`, and the EMA's
    #          restore() precedes the train() call that re-arms the check.
    def invalidate(self):
        """forget every packed / converted form (call after writing parameters through `.data`, e.g. an EMA swap)"""
        self._c.clear()
        for c in self._cast.values():
            c[1] = None  # the 16-bit copies keep their storage and are refreshed in place on the next p()
        self._sum = None

    def guard(self) -> bool:
        """compare a checksum (_checksum: three norms per tensor accumulated in fp64, fused multi-tensor launches, ONE host sync) of the module's
        parameters with the one taken at the previous call; on a difference drop every cached form.  Returns True when the cache was
        dropped."""
        cur = self._checksum()
        if cur is None:
            return False
        last = getattr(self, "_sum", None)
        # (no previous checksum: forms packed before the first guarded pass cannot be vouched for)
        changed = (bool(self._c) or bool(self._cast)) if last is None else (
            last.shape != cur.shape or last.device != cur.device or not torch.equal(last, cur))
        if changed:
            self.invalidate()
        self._sum = cur
        return changed

    def _checksum(self) -> Optional[torch.Tensor]:
        """per tensor: L1 and L2 norms and the L2 norm of (p + 1/2) -- the last one moves under sign flips, which the plain norms are
        blind to -- all ACCUMULATED IN FP64 whatever the parameters' dtype (in fp16 the L1 norm of a 512 x 512 x 27 conv weight
        overflows to inf, and inf == inf; a bf16 L2 norm carries 8 bits; an fp32 sum over 7 M elements hides a change of one)"""
        ps = [p.detach() for p in self.m.parameters()]
        if not ps:
            return None
        with torch.no_grad():
            f64 = torch.float64
            sums = (list(torch._foreach_norm(ps, 1, dtype=f64)) + list(torch._foreach_norm(ps, 2, dtype=f64)) +
                    list(torch._foreach_norm(torch._foreach_add(ps, 0.5), 2, dtype=f64)))
            return torch.stack(sums)

    def computing_in(self, dtype: Optional[torch.dtype]):
        """context manager: `compute_dtype` = dtype inside, the previous value afterwards (the dtype is per PASS, not per cache: a
        backward under autocast must not leave later export_packed / direct engine calls looking at the 16-bit copies)"""
        wc = self

        class _Ctx:
            def __enter__(self_):
                self_.prev = wc.compute_dtype
                wc.compute_dtype = dtype

            def __exit__(self_, *exc):
                wc.compute_dtype = self_.prev
                return False
        return _Ctx()

    def p(self, name: str) -> torch.nn.Parameter:
        """the module tree's parameter `name`, as nn.Module.get_parameter -- which walks the dotted path on every call (7 us each,
        ~4 lookups per conv launch: 2.7 ms of host time per encode + decode, more than the GPU time of an image-mode pass).  The
        path is resolved once to (the owning submodule's _parameters dict, leaf); the dict is read on every call, so a parameter
        replaced inside its submodule (load_state_dict, .to(), .cuda(), assignment) is always the current one."""
        hit = self._names.get(name)
        if hit is None:
            path, _, leaf = name.rpartition(".")
            sub = self.m.get_submodule(path) if path else self.m
            if sub._parameters.get(leaf) is None:
                raise AttributeError(f"{type(self.m).__name__} has no parameter {name}")
            hit = self._names[name] = (sub._parameters, leaf)
        par = hit[0].get(hit[1])
        if par is None:
            raise AttributeError(f"{type(self.m).__name__} has no parameter {name}")
        cd = self.compute_dtype
        if cd is not None and par.dtype == torch.float32 and par.dim() >= 2:
            mkey = (par.data_ptr(), par._version, par.device)
            c = self._cast.get(name)
            if c is None or c[0].dtype != cd or c[0].device != par.device or c[0].shape != par.shape:
                c = self._cast[name] = [par.detach().to(cd), mkey]
            elif c[1] != mkey:
                # in place: the copy keeps its storage and its _version moves on, so every packed form keyed on (data_ptr, _version)
                # is rebuilt -- a fresh tensor could land on the freed copy's address with _version 0 and look unchanged
                c[0].copy_(par.detach())
                c[1] = mkey
            return c[0]
        return par

    def act_bound(self, norm_pre: str, sigmas: float = 8.0) -> float:
        """upper bound of |SiLU(gamma n + beta)| for a normalised n within `sigmas`: sigmas max|gamma| + max|beta| (one host sync per
        GroupNorm, cached with its parameters).  Elements beyond it lose only their fp6 CORRECTION terms (include/cvvae.h)."""
        g = self.p(norm_pre + ".weight")
        b = self.p(norm_pre + ".bias")
        key = self._key(g, b)
        tag = norm_pre + "#bound"
        hit = self._c.get(tag)
        if hit is None or hit[0] != key:
            # ONE power of two scales every channel of the operand: channels far below the largest one would land in e3m2's
            # subnormals and lose their corrections, so a norm whose per-channel bounds spread over more than 3 binades (max > 8 x
            # median) returns 0.0 = "no bound": its convs keep the bf8 form, which needs no scale
            bc = (sigmas * g.detach().abs().float() + b.detach().abs().float()).flatten()
            top, mid = float(bc.max()), float(bc.median())
            hit = (key, max(top, 1e-6) if (top <= 8.0 * max(mid, 1e-30) and top <= self.fp6_bound_cap) else 0.0)
            self._c[tag] = hit
        return hit[1]

    def _key(self, *ps):
        return tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in ps if p is not None)

    def conv(self, pre: str, k: Tuple[int, int, int], cin_pad: Optional[int] = None, time_folds: bool = False,
             wscale: Optional[float] = None, act_norm: Optional[str] = None) -> ops.PackedConv:
        """time_folds (k = (3, kH, kW)): packed with the summed time slots for boundary frames (ops.pack_weight_tfolds);
        wscale (fp32 models): pack with this power-of-two scale (a fused shortcut shares its conv's accumulators);
        act_norm: the GroupNorm whose output (+ SiLU) this stride-1 conv consumes through its prologue -- fast fp32 models then take
        the fp6-correction form with the bound derived from that norm's affine"""
        w = self.p(pre + ".weight")
        b = self.p(pre + ".bias")
        key = self._key(w, b)
        q6 = (self.fast and self.fast6 and act_norm is not None and w.dtype == torch.float32 and wscale is None
              and tuple(k) in ((3, 3, 3), (1, 3, 3)) and self.act_bound(act_norm) > 0.0)
        fast = "fp6" if q6 else self.fast
        tag = (pre + "#tf" if time_folds else (pre if wscale is None else f"{pre}#ws{wscale}")) + ("#q6" if q6 else self._q())
        hit = self._c.get(tag)
        if hit is not None and hit[0] == key:
            if q6:
                hit[1].act_bound = self.act_bound(act_norm)
            return hit[1]
        taps = k[0] * k[1] * k[2]
        co, ci = w.shape[0], w.shape[1]
        assert w.numel() == co * ci * taps, f"{pre}: weight {tuple(w.shape)} is not a {k} kernel"
        if time_folds:
            pw = ops.pack_weight_tfolds(w.detach().reshape(co, ci, *k), b.detach(), cin_pad=cin_pad, fast=fast)
        else:
            pw = ops.pack_weight(w.detach().reshape(co, ci, taps), b.detach(), k, cin_pad=cin_pad, wscale=wscale, fast=fast)
        if q6:
            pw.act_bound = self.act_bound(act_norm)
        if wscale is not None:  # one scaled form per prefix: older '#ws<scale>' variants (another conv2 scale) are dead weight
            for t in [t for t in self._c if t.startswith(pre + "#ws") and t != tag]:
                del self._c[t]
        self._c[tag] = (key, pw, (pre + ".weight", pre + ".bias"))
        return pw

    def conv_dgrad(self, pre: str, k: Tuple[int, int, int], cin_pad: Optional[int] = None) -> ops.PackedConv:
        """The weights of the INPUT-GRADIENT convolution of a stride-1, zero-padded conv (or linear layer) `pre`: taps flipped,
        Cin and Cout exchanged, no bias -- conv(gy, conv_dgrad(pre)) with the forward's padding is autograd's grad_input
        (cvvae_amd/grad.py).  k: the forward kernel, (1,1,1) for nn.Linear / 1x1 weights."""
        w = self.p(pre + ".weight")
        key = self._key(w)
        tag = f"{pre}#dgrad"
        hit = self._c.get(tag)
        if hit is not None and hit[0] == key:
            return hit[1]
        taps = k[0] * k[1] * k[2]
        co, ci = w.shape[0], w.shape[1]
        assert w.numel() == co * ci * taps, f"{pre}: weight {tuple(w.shape)} is not a {k} kernel"
        wt = w.detach().reshape(co, ci, taps).flip(2).transpose(0, 1).contiguous()  # [ci, co, taps], taps reversed = flipped in every axis
        pw = ops.pack_weight(wt, None, k, cin_pad=cin_pad)
        self._c[tag] = (key, pw, (pre + ".weight",))
        return pw

    def conv_upfold(self, pre: str, tfold: int = 0, time_folds: bool = False, fp6: bool = False) -> ops.PackedConv:
        """Upsample3D conv weights folded into the four 3x2x2 (tfold: 1x2x2) phase kernels (ops.pack_weight_upfold).
        fp6 (fast fp32 models, 3x2x2 phases): the fp6-correction form -- the launch then needs a device-side bound of its operand
        (upsample_conv: the residual stream has no GroupNorm in front)"""
        w = self.p(pre + ".weight")
        b = self.p(pre + ".bias")
        key = self._key(w, b)
        fp6 = bool(fp6 and self.fast and self.fast6 and w.dtype == torch.float32 and not tfold)
        tag = f"{pre}#upfold{tfold}{'tf' if time_folds else ''}" + ("#q6" if fp6 else self._q())
        hit = self._c.get(tag)
        if hit is not None and hit[0] == key:
            return hit[1]
        pw = ops.pack_weight_upfold(w.detach(), b.detach(), tfold, time_folds=time_folds, fast="fp6" if fp6 else self.fast)
        self._c[tag] = (key, pw, (pre + ".weight", pre + ".bias"))
        return pw

    def conv_upfold2d(self, pre: str) -> ops.PackedConv:
        """Upsample2D conv weights [Cout, Cin, 3, 3] as the four folded 1x2x2 phase kernels: the 2-D weight is the centre time tap
        of an otherwise zero 3x3x3 weight (pack_weight_upfold tfold 2 = centre tap only)."""
        w = self.p(pre + ".weight")
        b = self.p(pre + ".bias")
        key = self._key(w, b)
        tag = f"{pre}#upfold2d" + self._q()
        hit = self._c.get(tag)
        if hit is not None and hit[0] == key:
            return hit[1]
        co, ci, kh, kw = w.shape
        assert (kh, kw) == (3, 3), f"{pre}: weight {tuple(w.shape)} is not a 3x3 kernel"
        w3 = torch.zeros((co, ci, 3, 3, 3), dtype=w.dtype, device=w.device)
        w3[:, :, 1] = w.detach()
        pw = ops.pack_weight_upfold(w3, b.detach(), 2, fast=self.fast)
        pw.alg_taps = 9
        self._c[tag] = (key, pw, (pre + ".weight", pre + ".bias"))
        return pw

    def conv_rowpack(self, pre: str, time_folds: bool = False) -> ops.PackedConv:
        """the networks' first layer ([Cout, 3, 3, 3, 3]) as the (3,3,1) conv over the row-packed input (ops.pack_weight_rowpack)"""
        w = self.p(pre + ".weight")
        b = self.p(pre + ".bias")
        key = self._key(w, b)
        tag = f"{pre}#rowpack{'tf' if time_folds else ''}"
        hit = self._c.get(tag)
        if hit is not None and hit[0] == key:
            return hit[1]
        pw = ops.pack_weight_rowpack(w.detach(), b.detach(), time_folds=time_folds)
        self._c[tag] = (key, pw, (pre + ".weight", pre + ".bias"))
        return pw

    def conv_tapsn(self, pre: str, time_folds: bool = False):
        """the decoders' last layer as the taps-in-N (3,1,1) conv (ops.pack_weight_tapsn) + its fp32 bias"""
        w = self.p(pre + ".weight")
        b = self.p(pre + ".bias")
        key = self._key(w, b)
        tag = f"{pre}#tapsn{'tf' if time_folds else ''}"
        hit = self._c.get(tag)
        if hit is None or hit[0] != key:
            hit = (key, ops.pack_weight_tapsn(w.detach(), time_folds=time_folds), (pre + ".weight", pre + ".bias"))
            self._c[tag] = hit
        bhit = self._c.get(pre + "#f32bias")  # (its own entry: the packed-weight file carries packed weights only)
        if bhit is None or bhit[0] != key:
            bhit = (key, b.detach().float().contiguous())
            self._c[pre + "#f32bias"] = bhit
        return hit[1], bhit[1]

    def conv_t1(self, pre: str, mode: str, cin_pad: Optional[int] = None) -> ops.PackedConv:
        """3 x kH x kW weights as the single-frame (T = 1) input sees them: time taps summed ('sum') or centre tap ('center')."""
        w = self.p(pre + ".weight")
        b = self.p(pre + ".bias")
        key = self._key(w, b)
        tag = f"{pre}#t1{mode}" + self._q()
        hit = self._c.get(tag)
        if hit is not None and hit[0] == key:
            return hit[1]
        pw = ops.pack_weight_t1(w.detach(), b.detach(), mode, cin_pad=cin_pad, fast=self.fast)
        self._c[tag] = (key, pw, (pre + ".weight", pre + ".bias"))
        return pw

    # ---- persistent packed-weight cache: the packed forms built so far, keyed by a fingerprint of their source
    #      parameters, so that a later process with the same checkpoint installs them instead of packing again
    @staticmethod
    def _fingerprint(p: torch.Tensor):
        f = p.detach().reshape(-1)  # (reductions accumulate in fp64 without materialising a converted copy)
        return (tuple(p.shape), str(p.dtype), float(f.sum(dtype=torch.float64)),
                float(torch.linalg.vector_norm(f, 1, dtype=torch.float64)), float(torch.linalg.vector_norm(f, 2, dtype=torch.float64)))

    def export_packed(self) -> dict:
        """{tag: {"names": source parameter names, "fp": their fingerprints, "pw": PackedConv fields}} of every packed conv weight
        in the cache (the forms a pass actually used: run the shapes of interest once before exporting)"""
        import dataclasses
        out = {}
        if self.compute_dtype is not None:  # (fingerprints and keys are those of the MASTER parameters, never of autocast copies)
            with self.computing_in(None):
                return self.export_packed()
        for tag, ent in self._c.items():
            if len(ent) < 3 or not dataclasses.is_dataclass(ent[1]):
                continue  # norm tables / summed biases: cheaper to rebuild than to read
            names = ent[2]
            # entries are refreshed lazily, on access: one packed BEFORE a load_state_dict / in-place edit still holds the OLD
            # weights, and exporting it under the fingerprint of the CURRENT parameters would hand a later process stale
            # weights that pass the import check.  Only entries whose key still matches the live parameters are exported.
            if ent[0] != self._key(*[self.p(n) for n in names]):
                continue
            out[tag] = {"names": list(names), "fp": [self._fingerprint(self.p(n)) for n in names],
                        "pw": {f.name: getattr(ent[1], f.name) for f in dataclasses.fields(ent[1])}}
        return out

    def import_packed(self, blob: dict) -> int:
        """install the exported entries whose source parameters still have the recorded fingerprints (shape, dtype, three
        moments) on this module; the rest is packed on demand as usual.  Returns the number installed."""
        n = 0
        if self.compute_dtype is not None:
            with self.computing_in(None):
                return self.import_packed(blob)
        fresh = not self._c and not self._cast
        for tag, e in blob.items():
            try:
                ps = [self.p(nm) for nm in e["names"]]
            except AttributeError:
                continue
            if [self._fingerprint(p) for p in ps] != [tuple(fp) if not isinstance(fp, tuple) else fp for fp in e["fp"]]:
                continue
            f = dict(e["pw"])
            dev = ps[0].device
            f["w"], f["bias"], f["k"] = f["w"].to(dev), f["bias"].to(dev), tuple(f["k"])
            self._c[tag] = (self._key(*ps), ops.PackedConv(**f), tuple(e["names"]))
            n += 1
        if fresh and n and getattr(self, "_sum", None) is None:
            # every entry was just checked against the parameters' fingerprints: the first guarded pass may keep them
            self._sum = self._checksum()
        return n

    def bias_sum(self, pre_a: str, pre_b: str) -> torch.Tensor:
        """fp32 b_a + b_b padded to a multiple of 32: the bias of a conv with a fused 1x1 shortcut"""
        ba, bb = self.p(pre_a + ".bias"), self.p(pre_b + ".bias")
        key = self._key(ba, bb)
        tag = f"{pre_a}+{pre_b}#bias"
        hit = self._c.get(tag)
        if hit is not None and hit[0] == key:
            return hit[1]
        out = torch.zeros(ops.round_up(ba.numel(), 32), dtype=torch.float32, device=ba.device)
        out[:ba.numel()] = ba.detach().float() + bb.detach().float()
        self._c[tag] = (key, out)
        return out

    def norm(self, pre: str) -> Tuple[torch.Tensor, torch.Tensor]:
        g = self.p(pre + ".weight")
        b = self.p(pre + ".bias")
        key = self._key(g, b)
        hit = self._c.get(pre)
        if hit is not None and hit[0] == key:
            return hit[1]
        val = (g.detach().float().contiguous(), b.detach().float().contiguous())
        self._c[pre] = (key, val)
        return val

    def has(self, name: str) -> bool:
        if name in self._absent:
            return False
        try:
            self.p(name)
            return True
        except AttributeError:
            self._absent.add(name)
            return False


def switches_key(wc=None) -> tuple:
    """every execution switch that changes the launch sequence of a pass: part of the hipGraph cache key (modeling._forward_graphed),
    so that flipping one in a live process re-captures instead of replaying the other setting's graph"""
    return (fold_upsample(), fold_t1(), fuse_shortcut(), fold_time(), rowpack_conv_in(), tapsn_conv_out(), fused_attention(),
            per_frame_stats_from_records(), os.environ.get("CVVAE_PREPASS", "auto"), os.environ.get("CVVAE_CONV_FORCE", ""),
            os.environ.get("CVVAE_F32_FP6_UPS", "1"),
            ops.four_wave(), None if wc is None else (wc.fast, wc.fast6, wc.fp6_bound_cap, str(wc.compute_dtype)))


def fold_upsample() -> bool:
    """Upsample3D = nearest x(1,2,2) + 3x3x3 conv.  Default: run it as four 3x2x2 phase convolutions over the stored input
    with folded weights (2.25x fewer MFMAs; differs from the 27-tap form only by one rounding of each folded weight).
    CVVAE_FOLD_UPSAMPLE=0 selects the 27-tap gather form (bit-for-bit the reference's summation terms)."""
    return os.environ.get("CVVAE_FOLD_UPSAMPLE", "1") != "0"


def fuse_shortcut() -> bool:
    """ResnetBlock3D with a channel change: conv2 and the 1x1 shortcut accumulate in the same MFMA registers (one launch;
    the shortcut tensor never goes through HBM).  CVVAE_FUSE_SHORTCUT=0 runs the shortcut as its own 1x1 conv launch."""
    return os.environ.get("CVVAE_FUSE_SHORTCUT", "1") != "0"


def prepass(x: torch.Tensor, k: Tuple[int, int, int], cout: int) -> bool:
    """GroupNorm+SiLU in front of a conv: fused into the conv's staging (prologue), or applied ONCE by cvvae_gn_silu_apply and
    the conv run without prologue (bit-identical results).  The fused form evaluates the activation for every halo copy and every
    N-tile of a pixel, and on gfx950 that VALU work adds to the MFMA time instead of hiding under it; the pass costs one read +
    write of the activation.  CVVAE_PREPASS: "0" never, "1" always, "k333" every 3x3x3 conv, default = the measured policy."""
    mode = os.environ.get("CVVAE_PREPASS", "auto")
    if mode == "0":
        return False
    if mode == "1":
        return True
    if mode == "k333":
        return k[0] == 3
    return PREPASS_AUTO(x, k, cout)


def PREPASS_AUTO(x, k, cout) -> bool:  # measured policy; round-2 default: fused everywhere
    return False


def _activated(x: torch.Tensor, kw: dict, k: Tuple[int, int, int], cout: int) -> Tuple[torch.Tensor, dict]:
    """(input, conv kwargs) with the GN+SiLU prologue either kept fused or applied by the pass (see prepass)"""
    if kw.get("prologue") == L.PRO_GN_SILU and not kw.get("gn_per_frame") and prepass(x, k, cout):
        return ops.gn_silu_apply(x, kw["gn"]), dict(kw, prologue=L.PRO_NONE, gn=None)
    return x, kw


def _shortcut_scale_fits(wc: WeightCache, sc_name: str, pw2, dtype) -> bool:
    """fp32 models: conv2's pack scale puts max |w_conv2| in [512, 1024); the fused 1x1 shortcut shares it.  Its fp16 hi part
    must stay finite (max |w_sc| * scale < 2^15 leaves headroom for the rounding); otherwise the block runs unfused."""
    if dtype != torch.float32:
        return True
    w = wc.p(sc_name + ".weight")
    key = wc._key(w)
    hit = wc._c.get(sc_name + "#absmax")
    if hit is None or hit[0] != key:
        hit = (key, float(w.detach().abs().max()))
        wc._c[sc_name + "#absmax"] = hit
    return hit[1] * pw2.wscale < 32768.0


def _fused_prologue(x: torch.Tensor, k: Tuple[int, int, int], cout: int) -> bool:
    """does THIS launch keep its GroupNorm + SiLU in the conv's staging?  The one decision (prepass) both the weight form (fp6
    corrections need the fused prologue: WeightCache.conv act_norm) and the launch follow."""
    return not prepass(x, k, cout)


def _cout(wc: "WeightCache", pre: str) -> int:
    return int(wc.p(pre + ".weight").shape[0])


def resnet_tail(wc: WeightCache, x: torch.Tensor, h: torch.Tensor, pre: str, sc_name: str, g2, want_stats: bool):
    """conv2 (per-frame 3x3 over GN+SiLU(h), zero pad) + shortcut(x) + add -- vae_blocks3d_sd3.py:559-567, vae_models.py:404-410."""
    fused = _fused_prologue(h, (1, 3, 3), _cout(wc, pre + ".conv2"))
    pw2 = wc.conv(pre + ".conv2", (1, 3, 3), act_norm=pre + ".norm2" if fused else None)
    kw = dict(pad=P2D, pad_mode_hw=ZERO, prologue=L.PRO_GN_SILU, gn=g2, gn_out=G32 if want_stats else 0)
    if not fused:
        h = ops.gn_silu_apply(h, g2)
        kw.update(prologue=L.PRO_NONE, gn=None)
    if not wc.has(sc_name + ".weight"):
        y = ops.conv(h, pw2, residual=x, **kw)
    elif fuse_shortcut() and pw2.dt not in (L.F32Q, L.F32Q6) and _shortcut_scale_fits(wc, sc_name, pw2, x.dtype):
        # (fp32 models: the shortcut weights are packed with conv2's power-of-two scale -- one accumulator set; the fast-fp32
        # kernels have no fused shortcut, and a shortcut whose weights would overflow fp16 under conv2's scale is not fused
        # either: the 1x1 then runs as its own three-MFMA launch with its own scale)
        pws = wc.conv(sc_name, (1, 1, 1), wscale=pw2.wscale if x.dtype == torch.float32 else None)
        y = ops.conv(h, pw2, shortcut=(x, pws), bias=wc.bias_sum(pre + ".conv2", sc_name), **kw)
    else:
        y = ops.conv(h, pw2, residual=conv1x1(wc, x, sc_name), **kw)
    return y if want_stats else (y, None)


def fold_t1() -> bool:
    """Single-frame inputs (image mode: T = 1 through the 3-D networks, e.g. the T2I pipeline's decode(z, num_frames=1)): the
    three time taps of every 3x3x3 conv read the same frame (replicate padding) or zeros (zero padding), so the layer runs as
    a per-frame conv with the coinciding taps summed into the weights (3x fewer MFMAs).  CVVAE_FOLD_T1=0 disables it."""
    return os.environ.get("CVVAE_FOLD_T1", "1") != "0"


def fold_time() -> bool:
    """Clips (T > 1): at the first / last frames two or three time taps of a 3x3x3 conv read the same stored frame (replicate time
    padding: CausalConv3d front 2, Conv3d replicate 1+1).  The packed weights then carry the summed slots W0+W1, W1+W2,
    W0+W1+W2 (cvvae_pack_weights_tfolds) and the kernel multiplies such a frame once: 6-20 % fewer MFMAs on the causal encoder
    convs, 4-13 % on the decoder's (T = 17 ... 5); results differ from the unfolded form only by the rounding of the summed
    weights.  (Zero time padding needs no weights: the kernel skips the padding frames, bit-exactly.)  CVVAE_FOLD_TIME=0 disables."""
    return os.environ.get("CVVAE_FOLD_TIME", "1") != "0"


def conv3(wc: WeightCache, x: torch.Tensor, pre: str, *, pad, pad_mode_t, pad_mode_hw, stride=(1, 1, 1), cin_pad=None,
          act_norm: Optional[str] = None, **kw):
    """One 3x3x3 convolution of the path (CausalConv3d / Conv3d / nn.Conv3d / Downsample3D).  On a single-frame input whose
    time padding makes the three taps coincide it runs as the temporally folded 1x3x3 conv (fold_t1).
    act_norm: name of the GroupNorm behind the GN+SiLU prologue (WeightCache.conv)."""
    if not (kw.get("prologue") == L.PRO_GN_SILU and tuple(stride) == (1, 1, 1) and not kw.get("gn_per_frame")
            and _fused_prologue(x, (3, 3, 3), _cout(wc, pre))):
        act_norm = None
    if x.shape[1] == 1 and fold_t1() and pad[0][0] + pad[0][1] == 2:
        pw = wc.conv_t1(pre, "sum" if pad_mode_t == REP else "center", cin_pad=cin_pad)
        x, kw = _activated(x, kw, (1, 3, 3), pw.cout)
        return ops.conv(x, pw, stride=(1, stride[1], stride[2]), pad=((0, 0), pad[1], pad[2]), pad_mode_hw=pad_mode_hw, **kw)
    tf = pad_mode_t == REP and fold_time()
    pw = wc.conv(pre, (3, 3, 3), cin_pad=cin_pad, time_folds=tf, act_norm=act_norm)
    x, kw = _activated(x, kw, (3, 3, 3), pw.cout)
    return ops.conv(x, pw, stride=stride, pad=pad, pad_mode_t=pad_mode_t, pad_mode_hw=pad_mode_hw, **kw)


def upsample_conv(wc: WeightCache, h: torch.Tensor, pre: str, pad, mode_t, mode_hw, up_time: bool):
    """Upsample3D.forward (vae_blocks3d_sd3.py:314-364, vae_models.py:214-235) in one launch: nearest-2x + conv + (up_time)
    channel->time shuffle and drop of frame 0; also returns the GroupNorm partials of the result."""
    om = L.OUT_TIME_SHUFFLE if up_time else L.OUT_NDHWC
    if h.shape[1] == 1 and fold_t1() and fold_upsample() and pad[0][0] + pad[0][1] == 2:
        return ops.conv(h, wc.conv_upfold(pre, 1 if mode_t == REP else 2), pad=((0, 0), pad[1], pad[2]), pad_mode_hw=mode_hw,
                        upsample2x=2, out_mode=om, gn_out=G32)
    if fold_upsample():
        # fast fp32 models: the fp6-correction form (1.5 instead of 2 MFMA units per tap), with the operand's bound taken on the
        # device -- one max-abs reduction over the input, the residual stream, which no GroupNorm bounds (CVVAE_F32_FP6_UPS=0: bf8)
        fp6 = (h.dtype == torch.float32 and wc.fast and wc.fast6 and wc.compute_dtype is None
               and os.environ.get("CVVAE_F32_FP6_UPS", "1") != "0")
        pw = wc.conv_upfold(pre, time_folds=mode_t == REP and fold_time(), fp6=fp6)
        bound = torch.linalg.vector_norm(h.reshape(-1), float("inf")).reshape(1) if pw.dt == L.F32Q6 else None
        return ops.conv(h, pw, pad=pad, pad_mode_t=mode_t, pad_mode_hw=mode_hw, upsample2x=2, out_mode=om, gn_out=G32,
                        act_bound_dev=bound)
    return ops.conv(h, wc.conv(pre, (3, 3, 3), time_folds=mode_t == REP and fold_time()), pad=pad, pad_mode_t=mode_t,
                    pad_mode_hw=mode_hw, upsample2x=True, out_mode=om, gn_out=G32)


def _flat(x: torch.Tensor) -> torch.Tensor:
    """[B,T,H,W,C] -> [B,1,1,T*H*W,C] view for the 1-D pixel tiles of the 1x1x1 conv (batch rows kept: GroupNorm
    statistics are per sample)."""
    return x.view(x.shape[0], 1, 1, -1, x.shape[-1])


def conv1x1(wc: WeightCache, x: torch.Tensor, pre: str, residual: Optional[torch.Tensor] = None, prologue=L.PRO_NONE,
            gn=None, gn_out: int = 0):
    """1x1(x1) conv / nn.Linear on the flattened pixels of every sample.  gn_out: also return the output's GroupNorm
    partials (per sample)."""
    pw = wc.conv(pre, (1, 1, 1))
    y = ops.conv(_flat(x), pw, prologue=prologue, gn=gn, residual=_flat(residual) if residual is not None else None,
                 gn_out=gn_out)
    if gn_out:
        y, part = y
        return y.view(*x.shape[:-1], pw.cout), part
    return y.view(*x.shape[:-1], pw.cout)


G32 = 32  # every GroupNorm of both families has 32 groups


def _norm(wc: WeightCache, x: torch.Tensor, part, pre: str, eps: float):
    """(scale, shift) of the 5-D GroupNorm `pre` applied to x: from the statistics its producer's epilogue emitted when
    available, else by a statistics pass over x."""
    if part is not None:
        return ops.gn_finalize(part, *wc.norm(pre), eps)
    return ops.gn_stats(x, *wc.norm(pre), eps)


# --------------------------------------------------------------------------------------------------------
# single-head spatial self-attention per frame (both families)
# --------------------------------------------------------------------------------------------------------
def spatial_attention(wc: WeightCache, x: torch.Tensor, norm: str, q: str, k: str, v: str, proj: str, eps: float,
                      residual: bool, gn_out: int = 0, tape: Optional[list] = None, xp=None):
    """sd3: AttentionWithExtraDim (vae_blocks3d_sd3.py:119-147) over diffusers Attention.
    vae3d: MemoryEfficientAttnBlock.attention + proj_out (vae_models.py:500-537).
    Per frame: GN(32) over (C/32, H*W) -> q,k,v (1x1) -> softmax(q k^T / sqrt(C)) v -> proj (+ x)."""
    B, T, H, W, C = x.shape
    N = H * W
    gamma, beta = wc.norm(norm)
    # rows = B*T: per-frame statistics -- merged from the producer's epilogue records when it was a per-frame conv over these
    # frames (xp.frames: the ResnetBlock tail in front of the attention), else by a statistics pass over x
    if xp is not None and getattr(xp, "frames", 0) == T and xp.slabs % T == 0 and per_frame_stats_from_records():
        gn = ops.gn_finalize(xp, gamma, beta, eps, frames=T)
    else:
        gn = ops.gn_stats(x, gamma, beta, eps, per_frame=True)
    xf = x.view(B * T, 1, 1, N, C)                                    # frames as batch: GN row = frame
    qq = ops.conv(xf, wc.conv(q, (1, 1, 1)), prologue=L.PRO_GN, gn=gn)
    kk = ops.conv(xf, wc.conv(k, (1, 1, 1)), prologue=L.PRO_GN, gn=gn)
    vv = ops.conv(xf, wc.conv(v, (1, 1, 1)), prologue=L.PRO_GN, gn=gn)
    scale = float(C) ** -0.5
    fused = fused_attention() and C == 512 and x.dtype in (torch.float16, torch.bfloat16)
    o = None
    if fused:
        # one launch: S = Q K^T, softmax, O = P V (csrc/attention_kernel.hip); scores and probabilities never reach HBM
        vt = ops.transpose(vv.view(B * T, N, C), ld_out=ops.round_up(N, 32))        # [BT, C, ldvt], zero padded
        o = ops.attention_d512(qq.view(B * T, N, C), kk.view(B * T, N, C), vt, N, scale)
    if o is None or tape is not None:
        # the five-launch form: every frame's K (and V^T) is a weight matrix of its own, packed in one launch and consumed by ONE
        # batched conv launch (cvvae_conv_desc.w_batch_stride) -- QK^T with fp32 scores, softmax, PV.  It is the forward of fp32
        # models, and it supplies the probabilities the input-gradient pass reads again (tape): a taped pass of a 16-bit model takes
        # its OUTPUT from the fused launch above, so that it equals the inference pass bit for bit
        npad = ops.round_up(N, 128)
        kp = ops.pack_weight_batched(kk.view(B * T, N, C), (1, 1, 1), cin_pad=C, strides=(C, 1, 0), cout=N, cin=C)
        s = ops.conv(qq, kp, out_f32=True, alpha=scale, cout_pad=npad)                              # [BT,1,1,N,npad] fp32
        p = ops.softmax_rows(s.view(B * T * N, npad), N, x.dtype)                                  # [BT*N, npad]
        if o is None:
            vt = ops.transpose(vv.view(B * T, N, C))                                               # [BT, C, N]
            vp = ops.pack_weight_batched(vt, (1, 1, 1), cin_pad=npad, strides=(N, 1, 0), cout=C, cin=N)
            o = ops.conv(p.view(B * T, 1, 1, N, npad), vp)                                         # [BT,1,1,N,C]
        if tape is not None:  # what the input-gradient pass (grad.py) needs again
            tape.append(dict(op="attn", x=x, qq=qq, kk=kk, vv=vv, p=p, o=o, gn=gn, names=(norm, q, k, v, proj), eps=eps,
                             residual=residual))
    return conv1x1(wc, o.view(B, T, H, W, C), proj, residual=x if residual else None, gn_out=gn_out)


def rowpack_conv_in() -> bool:
    """conv_in of the encoders (3 -> 128 channels, 3x3x3) on clips: run as a (3,3,1) convolution over a row-packed, W-padded copy of
    the input whose 16 virtual channels are the three kW taps x 4 channel slots (include/cvvae.h in_overlap): K = 144 per output
    instead of 432 with the channels padded 3 -> 16, and a 4x smaller converted input.  Same products, another summation order
    inside the fp32 accumulator.  CVVAE_ROWPACK_IN=0 keeps the channel-padded form."""
    return os.environ.get("CVVAE_ROWPACK_IN", "1") != "0"


def encoder_conv_in(wc: WeightCache, x: torch.Tensor, cfg: dict, dtype: torch.dtype, pad, mode_t, mode_hw):
    """conv_in of both encoders: x NCDHW (or the padded NDHWC clip of encode_frames_u8) -> (h NDHWC, GroupNorm partials)"""
    nd = bool(cfg.get("ndhwc_in"))
    cin = wc.p("conv_in.weight").shape[1]
    if (rowpack_conv_in() and x.shape[1 if nd else 2] > 1 and dtype in (torch.float16, torch.bfloat16) and cin <= 4
            and pad[2] == (1, 1) and (x.dtype == dtype or not nd)):
        # (the device-side pixel path hands over its padded NDHWC clip: same values, hence the same bits as the NCDHW entry)
        xr = ops.ndhwc_to_rowpack(x, cin, mode_hw) if nd else ops.ncdhw_to_rowpack(x, dtype, mode_hw)
        pw = wc.conv_rowpack("conv_in", time_folds=mode_t == REP and fold_time())
        return ops.conv(xr, pw, pad=(pad[0], pad[1], (0, 0)), pad_mode_t=mode_t, pad_mode_hw=mode_hw, gn_out=G32, row_packed=True)
    h, cpad = _encoder_input(x, cfg, dtype)
    return conv3(wc, h, "conv_in", cin_pad=cpad, pad=pad, pad_mode_t=mode_t, pad_mode_hw=mode_hw, gn_out=G32)


def tapsn_conv_out() -> bool:
    """conv_out of the decoders (128 -> 3 channels, 3x3x3): the nine spatial taps moved into the GEMM's N axis -- a (3,1,1) conv
    with 27 of 32 MFMA columns useful, no spatial halo (GroupNorm + SiLU applied 1.5x per element instead of 2.7x) and a ninth of the
    MFMAs, then a gather pass that sums the nine neighbours' columns in fp32 (include/cvvae.h cvvae_conv_out_gather).  Same
    products, another summation order.  CVVAE_TAPSN_OUT=0 keeps the 32-column form."""
    return os.environ.get("CVVAE_TAPSN_OUT", "1") != "0"


def decoder_conv_out(wc: WeightCache, h: torch.Tensor, g, pad, mode_t, mode_hw, u8: bool = False):
    """norm_out + SiLU + conv_out of both decoders -> pixels NCDHW (or, u8: the scripts' uint8 frames [T,H,W,3], one clip)"""
    w = wc.p("conv_out.weight")
    if (tapsn_conv_out() and 9 * w.shape[0] <= 32 and w.shape[0] == 3
            and pad[1] == (1, 1) and pad[2] == (1, 1) and h.shape[-1] % 32 == 0):
        pw, bias = wc.conv_tapsn("conv_out", time_folds=mode_t == REP and fold_time())
        v = ops.conv(h, pw, pad=(pad[0], (0, 0), (0, 0)), pad_mode_t=mode_t, pad_mode_hw=mode_hw, prologue=L.PRO_GN_SILU, gn=g,
                     out_f32=True)                                                   # [B,T,H,W,32] fp32
        return ops.conv_out_gather(v, w.shape[0], bias, mode_hw, h.dtype, u8=u8)
    y = conv3(wc, h, "conv_out", pad=pad, pad_mode_t=mode_t, pad_mode_hw=mode_hw, prologue=L.PRO_GN_SILU, gn=g, out_mode=L.OUT_NCDHW)
    return ops.ncdhw_to_frames_u8(y) if u8 else y


def fused_attention() -> bool:
    """the attention core (QK^T, softmax, PV) of 16-bit models with 512 channels as one launch (cvvae_attention_d512) instead of
    pack K / QK^T with fp32 scores / row softmax / pack V^T / PV.  CVVAE_FUSED_ATTENTION=0: the five-launch form (also what fp32
    models and the input-gradient tape use)."""
    return os.environ.get("CVVAE_FUSED_ATTENTION", "1") != "0"


def per_frame_stats_from_records() -> bool:
    """the attention blocks' per-frame GroupNorm statistics come from the records the preceding ResnetBlock tail already wrote
    (cvvae_gn_finalize_frames) instead of a pass over the tensor.  CVVAE_FRAME_STATS_FROM_RECORDS=0: always the pass."""
    return os.environ.get("CVVAE_FRAME_STATS_FROM_RECORDS", "1") != "0"


def _encoder_input(x: torch.Tensor, cfg: dict, dtype: torch.dtype):
    """the encoder's NDHWC input, channel-padded for conv_in's K chunk: converted from the caller's NCDHW clip, or -- cfg
    "ndhwc_in" (the device-side pixel pre-processing, modeling.encode_frames_u8) -- the padded NDHWC clip as it arrives."""
    if cfg.get("ndhwc_in"):
        T, cs = x.shape[1], x.shape[-1]
        cpad = 32 if (T == 1 and fold_t1()) else 16
        if x.dtype != dtype or cs < cpad or not x.is_contiguous():
            raise ValueError(f"ndhwc_in: expected a contiguous [B,T,H,W,>={cpad}] {dtype} clip with zero pad channels")
        return x, cpad
    cpad = 32 if (x.shape[2] == 1 and fold_t1()) else 16  # the folded single-frame path runs 32-channel K-chunks
    return ops.ncdhw_to_ndhwc(x, cpad, dtype), cpad


# --------------------------------------------------------------------------------------------------------
# vae3d_sd3 family
# --------------------------------------------------------------------------------------------------------
def sd3_resnet(wc: WeightCache, x: torch.Tensor, xp, pre: str, causal: bool, want_stats: bool = True, tape: Optional[list] = None):
    """ResnetBlock3D.forward, vae_blocks3d_sd3.py:517-569: GN(eps 1e-6)+SiLU fused into conv1 (replicate pad, causal
    T(2,0) or (1,1)) and into conv2 (per-frame 3x3, zero pad); 1x1 shortcut; residual add in conv2's epilogue.
    xp: GroupNorm partials of x from its producer (or None).  Returns (out, partials of out or None)."""
    g1 = _norm(wc, x, xp, pre + ".norm1", 1e-6)
    h, hp = conv3(wc, x, pre + ".conv1", pad=PC if causal else P1, pad_mode_t=REP, pad_mode_hw=REP,
                     prologue=L.PRO_GN_SILU, gn=g1, gn_out=G32, act_norm=pre + ".norm1")
    g2 = ops.gn_finalize(hp, *wc.norm(pre + ".norm2"), 1e-6)
    if tape is not None:  # what the training-side backward (grad3d.py) reads again: block input, conv1 output, their statistics
        tape.append(dict(op="resnet3d", pre=pre, x=x, xp=xp, h=h, hp=hp, g1=g1, g2=g2, pad=PC if causal else P1, mode_t=REP, mode_hw=REP,
                         eps=1e-6, sc=".conv_shortcut"))
    return resnet_tail(wc, x, h, pre, pre + ".conv_shortcut", g2, want_stats)


def sd3_mid(wc: WeightCache, x: torch.Tensor, xp, pre: str, causal: bool, attention: bool, tape: Optional[list] = None):
    """UNetMidBlock3D.forward, vae_blocks3d_sd3.py:847-856."""
    x, xp = sd3_resnet(wc, x, xp, pre + ".resnets.0", causal, tape=tape)  # (its records also give the attention's per-frame statistics)
    if attention:
        a = pre + ".attentions.0"
        x, xp = spatial_attention(wc, x, a + ".group_norm", a + ".to_q", a + ".to_k", a + ".to_v", a + ".to_out.0", 1e-6, True,
                                  gn_out=G32, xp=xp, tape=tape)
    return sd3_resnet(wc, x, xp, pre + ".resnets.1", causal, tape=tape)


def sd3_encoder(wc: WeightCache, x: torch.Tensor, cfg: dict, tape: Optional[list] = None) -> torch.Tensor:
    """Encoder3D.forward, vae_models3d_sd3.py:162-208.  x: NCDHW (any float dtype) -> moments NCDHW.
    tape: a list that receives what the backward pass of the TRAINABLE encoder (grad3d.sd3_encoder_backward) reads again; the
    launches are the inference pass's own (same bits)."""
    dtype = wc.p("conv_in.weight").dtype
    causal = cfg["causal"]
    boc = cfg["block_out_channels"]
    pad = PC if causal else P1
    h, hp = encoder_conv_in(wc, x, cfg, dtype, pad, REP, REP)
    if tape is not None:
        tape.append(dict(op="conv_in", x=x, pad=pad, mode_t=REP, mode_hw=REP, ndhwc_in=bool(cfg.get("ndhwc_in"))))
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"]):
            h, hp = sd3_resnet(wc, h, hp, f"down_blocks.{i}.resnets.{j}", causal, tape=tape)
        if i != len(boc) - 1:  # Downsample3D vae_blocks3d_sd3.py:224-239; time stride on even blocks (:115)
            st = (2, 2, 2) if i % 2 == 0 else (1, 2, 2)
            if tape is not None:
                tape.append(dict(op="down3d", pre=f"down_blocks.{i}.downsamplers.0.conv", x=h, stride=st, pad=pad, mode_t=REP, mode_hw=REP))
            h, hp = conv3(wc, h, f"down_blocks.{i}.downsamplers.0.conv", stride=st,
                             pad=pad, pad_mode_t=REP, pad_mode_hw=REP, gn_out=G32)
    h, hp = sd3_mid(wc, h, hp, "mid_block", causal, cfg["mid_block_add_attention"], tape=tape)
    g = _norm(wc, h, hp, "conv_norm_out", 1e-6)
    if tape is not None:
        tape.append(dict(op="out3d", x=h, xp=hp, g=g, pad=pad, mode_t=REP, mode_hw=REP, eps=1e-6, norm="conv_norm_out"))
    return conv3(wc, h, "conv_out", pad=pad, pad_mode_t=REP, pad_mode_hw=REP,
                    prologue=L.PRO_GN_SILU, gn=g, out_mode=L.OUT_NCDHW)


def sd3_decoder(wc: WeightCache, z: torch.Tensor, cfg: dict, tape: Optional[list] = None) -> torch.Tensor:
    """Decoder3D.forward, vae_models3d_sd3.py:323-388.  z: NCDHW latents -> pixels NCDHW.
    tape: receives what grad3d.sd3_decoder_backward reads again (training the decoder; the launches are the inference pass's)."""
    dtype = wc.p("conv_in.weight").dtype
    causal = cfg["causal"]
    boc = cfg["block_out_channels"]
    pad = PC if causal else P1
    zin = z.shape[1]
    cpad = ops.round_up(zin, 32 if (z.shape[2] == 1 and fold_t1()) else 16)
    h = ops.ncdhw_to_ndhwc(z, cpad, dtype)
    if tape is not None:
        tape.append(dict(op="dec_in", x=h, pad=pad, mode_t=REP, mode_hw=REP, zin=zin))
    h, hp = conv3(wc, h, "conv_in", cin_pad=cpad, pad=pad, pad_mode_t=REP, pad_mode_hw=REP,
                     gn_out=G32)
    h, hp = sd3_mid(wc, h, hp, "mid_block", causal, cfg["mid_block_add_attention"], tape=tape)
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"] + 1):
            h, hp = sd3_resnet(wc, h, hp, f"up_blocks.{i}.resnets.{j}", causal, tape=tape)
        if i != len(boc) - 1:  # Upsample3D vae_blocks3d_sd3.py:314-364; up_time on even blocks (vae_models3d_sd3.py:289)
            up_time = i % 2 == 0
            if tape is not None:
                tape.append(dict(op="up3d", pre=f"up_blocks.{i}.upsamplers.0.conv", x=h, pad=pad, mode_t=REP, mode_hw=REP, up_time=up_time))
            h, hp = upsample_conv(wc, h, f"up_blocks.{i}.upsamplers.0.conv", pad, REP, REP, up_time)
    g = _norm(wc, h, hp, "conv_norm_out", 1e-6)
    if tape is not None:
        tape.append(dict(op="out3d", x=h, xp=hp, g=g, pad=pad, mode_t=REP, mode_hw=REP, eps=1e-6, norm="conv_norm_out"))
    return decoder_conv_out(wc, h, g, pad, REP, REP, u8=bool(cfg.get("u8_out")))


# --------------------------------------------------------------------------------------------------------
# the frozen 2-D "constraint" decoder of the training path (SD3 image VAE decoder per frame)
# --------------------------------------------------------------------------------------------------------
def c2d_resnet(wc: WeightCache, x: torch.Tensor, xp, pre: str, want_stats: bool = True, tape: Optional[list] = None,
               sc: str = ".conv_shortcut"):
    """ResnetBlock2D.forward, lvdm/modules/diffusionmodules/vae_blocks_sd3.py:368-421, on [frames,1,H,W,C]: GN(eps 1e-6)+SiLU
    fused into conv1 and conv2 (per-frame 3x3, zero pad), 1x1 shortcut and residual add in conv2's launch."""
    g1 = _norm(wc, x, xp, pre + ".norm1", 1e-6)
    # (this launch always keeps its prologue fused: no prepass decision in front of it)
    h, hp = ops.conv(x, wc.conv(pre + ".conv1", (1, 3, 3), act_norm=pre + ".norm1"), pad=P2D,
                     pad_mode_hw=ZERO, prologue=L.PRO_GN_SILU, gn=g1, gn_out=G32)
    g2 = ops.gn_finalize(hp, *wc.norm(pre + ".norm2"), 1e-6)
    if tape is not None:
        tape.append(dict(op="resnet", pre=pre, x=x, xp=xp, h=h, hp=hp))
    return resnet_tail(wc, x, h, pre, pre + sc, g2, want_stats)


def constraint_decoder2d(wc: WeightCache, z: torch.Tensor, cfg: dict, tape: Optional[list] = None) -> torch.Tensor:
    """DecoderWith3DWrapper.forward over Decoder.forward (lvdm/modules/diffusionmodules/vae_models_sd3.py:297-362, 390-398):
    z NCDHW [b,c,t,h,w] -> pixels [b,3,t,8h,8w], every frame decoded on its own ("b c t h w -> (b t) c h w": frames are the
    batch rows here, so every GroupNorm is per frame as in the reference).  tape: a list that receives what the input-gradient
    pass of the frozen decoder (grad.constraint_decoder2d_backward) reads again."""
    dtype = wc.p("conv_in.weight").dtype
    boc = cfg["block_out_channels"]
    B, zin, T = z.shape[0], z.shape[1], z.shape[2]
    cpad = ops.round_up(zin, 32)
    h = ops.ncdhw_to_ndhwc(z, cpad, dtype)
    h = h.view(B * T, 1, h.shape[2], h.shape[3], cpad)
    h, hp = ops.conv(h, wc.conv("conv_in", (1, 3, 3), cin_pad=cpad), pad=P2D, pad_mode_hw=ZERO, gn_out=G32)
    attn = cfg["mid_block_add_attention"]
    h, hp = c2d_resnet(wc, h, hp, "mid_block.resnets.0", want_stats=not attn, tape=tape)  # UNetMidBlock2D.forward, vae_blocks_sd3.py:669-681
    if attn:
        a = "mid_block.attentions.0"
        h, hp = spatial_attention(wc, h, a + ".group_norm", a + ".to_q", a + ".to_k", a + ".to_v", a + ".to_out.0", 1e-6, True,
                                  gn_out=G32, tape=tape)
    h, hp = c2d_resnet(wc, h, hp, "mid_block.resnets.1", tape=tape)
    for i in range(len(boc)):  # UpDecoderBlock2D.forward, vae_blocks_sd3.py:536-547
        for j in range(cfg["layers_per_block"] + 1):
            h, hp = c2d_resnet(wc, h, hp, f"up_blocks.{i}.resnets.{j}", tape=tape)
        if i != len(boc) - 1:  # Upsample2D.forward :178-230: nearest x2 + conv 3x3 (zero pad), as four folded 1x2x2 phase convs
            if tape is not None:
                tape.append(dict(op="up", pre=f"up_blocks.{i}.upsamplers.0.conv"))
            h, hp = ops.conv(h, wc.conv_upfold2d(f"up_blocks.{i}.upsamplers.0.conv"), pad=P2D, pad_mode_hw=ZERO, upsample2x=2,
                             gn_out=G32)
    g = _norm(wc, h, hp, "conv_norm_out", 1e-6)
    if tape is not None:
        tape.append(dict(op="out", x=h, xp=hp, B=B, T=T, zin=zin))
    y = ops.conv(h, wc.conv("conv_out", (1, 3, 3)), pad=P2D, pad_mode_hw=ZERO, prologue=L.PRO_GN_SILU, gn=g,
                 out_mode=L.OUT_NCDHW)                      # [b*t, 3, 1, H, W]
    return y.view(B, T, y.shape[1], y.shape[3], y.shape[4]).transpose(1, 2).contiguous()


# --------------------------------------------------------------------------------------------------------
# SURVEY 8(f) rank 4, second half: the frozen 2-D encoder / decoder of the SD2.1-compatible family (LDM layout) per frame
# --------------------------------------------------------------------------------------------------------
def _frames_in(x: torch.Tensor, cpad: int, dtype: torch.dtype) -> torch.Tensor:
    """[b,c,t,h,w] -> NDHWC [(b t),1,h,w,cpad]: 'b c t h w -> (b t) c h w' (frames are the batch rows: every GroupNorm is per frame)"""
    h = ops.ncdhw_to_ndhwc(x, cpad, dtype)
    return h.view(h.shape[0] * h.shape[1], 1, h.shape[2], h.shape[3], cpad)


def _frames_out(y: torch.Tensor, B: int, T: int) -> torch.Tensor:
    """NCDHW [(b t),c,1,h,w] -> [b,c,t,h,w]"""
    return y.view(B, T, y.shape[1], y.shape[3], y.shape[4]).transpose(1, 2).contiguous()


def ldm2d_encoder(wc: WeightCache, x: torch.Tensor, cfg: dict) -> torch.Tensor:
    """EncoderWith3DWrapper.forward over Encoder.forward (lvdm/modules/diffusionmodules/model.py:587-614, 877-887): every frame
    through conv_in -> levels of ResnetBlocks (GroupNorm eps 1e-6 + swish + 3x3, 1x1 nin_shortcut) with Downsample (zero pad
    right / bottom, 3x3 stride 2) -> mid (block_1, single-head attn_1, block_2) -> norm_out + swish + conv_out -> quant_conv 1x1.
    x NCDHW [b,c,t,h,w] -> moments [b,2z,t,h/f,w/f]."""
    dtype = wc.p("conv_in.weight").dtype
    B, T = x.shape[0], x.shape[2]
    nlev = len(cfg["ch_mult"])
    h = _frames_in(x, 32, dtype)
    h, hp = ops.conv(h, wc.conv("conv_in", (1, 3, 3), cin_pad=32), pad=P2D, pad_mode_hw=ZERO, gn_out=G32)
    for lvl in range(nlev):
        for j in range(cfg["num_res_blocks"]):
            h, hp = c2d_resnet(wc, h, hp, f"down.{lvl}.block.{j}", sc=".nin_shortcut")
        if lvl != nlev - 1:  # Downsample: F.pad (0,1,0,1) zeros, conv 3x3 stride 2 pad 0 (model.py:88-95)
            h, hp = ops.conv(h, wc.conv(f"down.{lvl}.downsample.conv", (1, 3, 3)), stride=(1, 2, 2), pad=((0, 0), (0, 1), (0, 1)),
                             pad_mode_hw=ZERO, gn_out=G32)
    h, hp = c2d_resnet(wc, h, hp, "mid.block_1", sc=".nin_shortcut")
    a = "mid.attn_1"
    h, hp = spatial_attention(wc, h, a + ".norm", a + ".q", a + ".k", a + ".v", a + ".proj_out", 1e-6, True, gn_out=G32, xp=hp)
    h, hp = c2d_resnet(wc, h, hp, "mid.block_2", sc=".nin_shortcut")
    g = _norm(wc, h, hp, "norm_out", 1e-6)
    zc = wc.p("conv_out.weight").shape[0]
    m = ops.conv(h, wc.conv("conv_out", (1, 3, 3)), pad=P2D, pad_mode_hw=ZERO, prologue=L.PRO_GN_SILU, gn=g,
                 cout_pad=ops.round_up(zc, 128) if wc.has("quant_conv.weight") else None,
                 out_mode=L.OUT_NDHWC if wc.has("quant_conv.weight") else L.OUT_NCDHW)
    if wc.has("quant_conv.weight"):  # legacy=True: nn.Conv2d(2z, 2z, 1) on the moments (model.py:870-885)
        pq = wc.conv("quant_conv", (1, 1, 1), cin_pad=m.shape[-1])
        q = ops.conv(_flat(m), pq).view(*m.shape[:-1], pq.cout)
        m = ops.ndhwc_to_ncdhw(q, pq.cout)
    return _frames_out(m, B, T)


def ldm2d_decoder(wc: WeightCache, z: torch.Tensor, cfg: dict) -> torch.Tensor:
    """DecoderWith3DWrapper.forward over Decoder.forward (model.py:728-772, 820-830): post_quant_conv 1x1 -> conv_in -> mid ->
    levels (num_res_blocks + 1 ResnetBlocks, Upsample = nearest x2 + 3x3) from the coarsest -> norm_out + swish + conv_out.
    z NCDHW [b,zc,t,h,w] -> pixels [b,out_ch,t,f*h,f*w]."""
    dtype = wc.p("conv_in.weight").dtype
    B, T = z.shape[0], z.shape[2]
    nlev = len(cfg["ch_mult"])
    if wc.has("post_quant_conv.weight"):
        h = _frames_in(z, 128, dtype)
        pq = wc.conv("post_quant_conv", (1, 1, 1), cin_pad=128)
        h = ops.conv(_flat(h), pq, cout_pad=32).view(*h.shape[:-1], 32)
    else:
        h = _frames_in(z, 32, dtype)
    h, hp = ops.conv(h, wc.conv("conv_in", (1, 3, 3), cin_pad=32), pad=P2D, pad_mode_hw=ZERO, gn_out=G32)
    h, hp = c2d_resnet(wc, h, hp, "mid.block_1", sc=".nin_shortcut")
    a = "mid.attn_1"
    h, hp = spatial_attention(wc, h, a + ".norm", a + ".q", a + ".k", a + ".v", a + ".proj_out", 1e-6, True, gn_out=G32, xp=hp)
    h, hp = c2d_resnet(wc, h, hp, "mid.block_2", sc=".nin_shortcut")
    for lvl in reversed(range(nlev)):
        for j in range(cfg["num_res_blocks"] + 1):
            h, hp = c2d_resnet(wc, h, hp, f"up.{lvl}.block.{j}", sc=".nin_shortcut")
        if lvl != 0:  # Upsample: nearest x2 + conv 3x3 pad 1 (model.py:68-75), as four folded 1x2x2 phase convs
            h, hp = ops.conv(h, wc.conv_upfold2d(f"up.{lvl}.upsample.conv"), pad=P2D, pad_mode_hw=ZERO, upsample2x=2, gn_out=G32)
    g = _norm(wc, h, hp, "norm_out", 1e-6)
    y = ops.conv(h, wc.conv("conv_out", (1, 3, 3)), pad=P2D, pad_mode_hw=ZERO, prologue=L.PRO_GN_SILU, gn=g, out_mode=L.OUT_NCDHW)
    return _frames_out(y, B, T)


# --------------------------------------------------------------------------------------------------------
# vae3d family
# --------------------------------------------------------------------------------------------------------
def _v3_pad(causal: bool):
    # CausalConv3d (vae_models.py:298-328): zero pad W,H; replicate T front 2.  nn.Conv3d(padding=1): zero everywhere.
    return (PC, REP, ZERO) if causal else (P1, ZERO, ZERO)


def v3_resnet(wc: WeightCache, x: torch.Tensor, xp, pre: str, causal: bool, want_stats: bool = True, tape: Optional[list] = None):
    """ResnetBlock3D.forward, vae_models.py:390-410 (GN eps 1e-5, swish, nin_shortcut 1x1x1).
    xp: GroupNorm partials of x from its producer (or None).  Returns (out, partials of out or None)."""
    pad, mt, mhw = _v3_pad(causal)
    g1 = _norm(wc, x, xp, pre + ".norm1", 1e-5)
    h, hp = conv3(wc, x, pre + ".conv1", pad=pad, pad_mode_t=mt, pad_mode_hw=mhw, prologue=L.PRO_GN_SILU,
                     gn=g1, gn_out=G32, act_norm=pre + ".norm1")
    g2 = ops.gn_finalize(hp, *wc.norm(pre + ".norm2"), 1e-5)
    if tape is not None:
        tape.append(dict(op="resnet3d", pre=pre, x=x, xp=xp, h=h, hp=hp, g1=g1, g2=g2, pad=pad, mode_t=mt, mode_hw=mhw, eps=1e-5,
                         sc=".nin_shortcut"))
    return resnet_tail(wc, x, h, pre, pre + ".nin_shortcut", g2, want_stats)


def v3_encoder(wc: WeightCache, x: torch.Tensor, cfg: dict, tape: Optional[list] = None) -> torch.Tensor:
    """Encoder.forward, vae_models.py:790-823.  tape: see sd3_encoder (grad3d.py trains this network too)."""
    dtype = wc.p("conv_in.weight").dtype
    causal = cfg["causal"]
    pad, mt, mhw = _v3_pad(causal)
    nlev = len(cfg["ch_mult"])
    h, hp = encoder_conv_in(wc, x, cfg, dtype, pad, mt, mhw)
    if tape is not None:
        tape.append(dict(op="conv_in", x=x, pad=pad, mode_t=mt, mode_hw=mhw, ndhwc_in=bool(cfg.get("ndhwc_in"))))
    for lvl in range(nlev):
        for j in range(cfg["num_res_blocks"]):
            h, hp = v3_resnet(wc, h, hp, f"down.{lvl}.block.{j}", causal, tape=tape)
        if lvl != nlev - 1:  # Downsample3D vae_models.py:251-263: zero pad right/bottom, replicate T front 2
            st = (2, 2, 2) if lvl % 2 == 0 else (1, 2, 2)
            dpad = ((2, 0), (0, 1), (0, 1))
            if tape is not None:
                tape.append(dict(op="down3d", pre=f"down.{lvl}.downsample.conv", x=h, stride=st, pad=dpad, mode_t=REP, mode_hw=ZERO))
            h, hp = conv3(wc, h, f"down.{lvl}.downsample.conv", stride=st, pad=dpad,
                             pad_mode_t=REP, pad_mode_hw=ZERO, gn_out=G32)
    h, hp = v3_resnet(wc, h, hp, "mid.block_1", causal, tape=tape)
    a = "mid.attn_1"
    h, hp = spatial_attention(wc, h, a + ".norm", a + ".q", a + ".k", a + ".v", a + ".proj_out", 1e-5, True, gn_out=G32, xp=hp,
                              tape=tape)
    h, hp = v3_resnet(wc, h, hp, "mid.block_2", causal, tape=tape)
    g = _norm(wc, h, hp, "norm_out", 1e-5)
    if tape is not None:
        tape.append(dict(op="out3d", x=h, xp=hp, g=g, pad=pad, mode_t=mt, mode_hw=mhw, eps=1e-5, norm="norm_out"))
    return conv3(wc, h, "conv_out", pad=pad, pad_mode_t=mt, pad_mode_hw=mhw, prologue=L.PRO_GN_SILU, gn=g,
                    out_mode=L.OUT_NCDHW)


def v3_attn_spatial_temporal(wc: WeightCache, x: torch.Tensor, a: str, xp=None, tape: Optional[list] = None):
    """MemoryEfficientAttnVideoBlock.forward, vae_models.py:619-629: spatial attention without residual, then over T
    per pixel: LayerNorm -> q_t,k_t,v_t -> attention -> proj_out_t; one residual.  Stays NDHWC throughout.
    Returns (out, GroupNorm partials of out)."""
    h = spatial_attention(wc, x, a + ".norm", a + ".q", a + ".k", a + ".v", a + ".proj_out", 1e-5, False, xp=xp, tape=tape)
    n = ops.layernorm(h, *wc.norm(a + ".norm_t"), 1e-5)
    q = conv1x1(wc, n, a + ".q_t")
    k = conv1x1(wc, n, a + ".k_t")
    v = conv1x1(wc, n, a + ".v_t")
    o = ops.temporal_attention(q, k, v)
    if tape is not None:  # (after the spatial part's own entry: the backward walks the tape in reverse)
        tape.append(dict(op="attn_t", pre=a, h=h, n=n, q=q, k=k, v=v, o=o))
    return conv1x1(wc, o, a + ".proj_out_t", residual=x, gn_out=G32)


def v3_decoder(wc: WeightCache, z: torch.Tensor, cfg: dict, tape: Optional[list] = None) -> torch.Tensor:
    """Decoder.forward, vae_models.py:960-1002.  tape: see sd3_decoder."""
    dtype = wc.p("conv_in.weight").dtype
    causal = cfg["causal"]
    pad, mt, mhw = _v3_pad(causal)
    nlev = len(cfg["ch_mult"])
    zin = z.shape[1]
    cpad = ops.round_up(zin, 32 if (z.shape[2] == 1 and fold_t1()) else 16)
    h = ops.ncdhw_to_ndhwc(z, cpad, dtype)
    if tape is not None:
        tape.append(dict(op="dec_in", x=h, pad=pad, mode_t=mt, mode_hw=mhw, zin=zin))
    h, hp = conv3(wc, h, "conv_in", cin_pad=cpad, pad=pad, pad_mode_t=mt, pad_mode_hw=mhw,
                     gn_out=G32)
    h, hp = v3_resnet(wc, h, hp, "mid.block_1", causal, tape=tape)
    h, hp = v3_attn_spatial_temporal(wc, h, "mid.attn_1", xp=hp, tape=tape)
    h, hp = v3_resnet(wc, h, hp, "mid.block_2", causal, tape=tape)
    for lvl in reversed(range(nlev)):
        for j in range(cfg["num_res_blocks"] + 1):
            h, hp = v3_resnet(wc, h, hp, f"up.{lvl}.block.{j}", causal, tape=tape)
        if lvl != 0:  # Upsample3D vae_models.py:214-235 (built non-causal, :936): zero pad W,H, replicate T (1,1)
            up_time = lvl % 2 == 1
            if tape is not None:
                tape.append(dict(op="up3d", pre=f"up.{lvl}.upsample.conv", x=h, pad=P1, mode_t=REP, mode_hw=ZERO, up_time=up_time))
            h, hp = upsample_conv(wc, h, f"up.{lvl}.upsample.conv", P1, REP, ZERO, up_time)
    g = _norm(wc, h, hp, "norm_out", 1e-5)
    if tape is not None:
        tape.append(dict(op="out3d", x=h, xp=hp, g=g, pad=pad, mode_t=mt, mode_hw=mhw, eps=1e-5, norm="norm_out"))
    return decoder_conv_out(wc, h, g, pad, mt, mhw, u8=bool(cfg.get("u8_out")))
