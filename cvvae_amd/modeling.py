"""Drop-in boundary: `CVVAEModel` / `CVVAESD3Model` with the reference's public API
(/root/reference/models/modeling_vae.py:20-341 and :344-667) over the MI355X HIP engine.

Same class names, constructor/config keys, `from_pretrained(path, subfolder=, torch_dtype=)`, `.encode(x).latent_dist`,
`.decode(z, num_frames=).sample`, `.forward`, `.encoder` / `.decoder` callables on NCDHW tensors, the same temporal
windows / spatial tiles / in-place blending, and the same state-dict key names and shapes, so
cvvae_inference_video.py and cvvae_sd3_inference_video.py run unchanged on top of `models/modeling_vae.py`.
diffusers is not a dependency: config / weight loading and the two small output types are implemented here.
"""
import json
import math
import os
import warnings
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from . import engine, ops


# ------------------------------------------------------------------------------------------------------
# parameter holders: they only own tensors under the reference's names; compute happens in engine.py
# ------------------------------------------------------------------------------------------------------
class ConvP(nn.Module):
    def __init__(self, cin: int, cout: int, ksize: Tuple[int, ...]):
        super().__init__()
        self.weight = nn.Parameter(torch.empty((cout, cin) + tuple(ksize)))
        self.bias = nn.Parameter(torch.empty(cout))
        fan_in = cin * int(math.prod(ksize)) if ksize else cin
        bound = 1.0 / math.sqrt(fan_in)  # torch's default conv/linear init (kaiming_uniform(a=sqrt(5)))
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            self.bias.uniform_(-bound, bound)


class NormP(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


def _holder(**children) -> nn.Module:
    m = nn.Module()
    for k, v in children.items():
        m.add_module(k, v)
    return m


def _resnet(cin: int, cout: int, shortcut_name: str, shortcut_k: Tuple[int, ...]) -> nn.Module:
    m = _holder(norm1=NormP(cin), conv1=ConvP(cin, cout, (3, 3, 3)), norm2=NormP(cout), conv2=ConvP(cout, cout, (3, 3)))
    if cin != cout:
        m.add_module(shortcut_name, ConvP(cin, cout, shortcut_k))
    return m


def _autocast_dtype(x: torch.Tensor) -> Optional[torch.dtype]:
    """the active GPU autocast dtype (float16 / bfloat16), or None outside torch.autocast"""
    try:
        on = torch.is_autocast_enabled("cuda")
    except TypeError:  # older signature
        on = torch.is_autocast_enabled()
    if not on:
        return None
    try:
        dt = torch.get_autocast_dtype("cuda")
    except AttributeError:
        dt = torch.get_autocast_gpu_dtype()
    return dt if dt in (torch.float16, torch.bfloat16) else None


class _Net(nn.Module):
    """Base of the four encoder/decoder modules: parameters + a WeightCache + `forward` on NCDHW tensors."""

    _program = None

    def __init__(self):
        super().__init__()
        self._wc = None
        self._cfg = {}
        self._unsupported = None  # set by the wrapper when the configuration cannot run on the kernels (message)
        self._graphs = None  # enable_hip_graphs(): {(shape, dtype, device, switches): _GraphEntry}
        self._graph_cap = 0
        self._fp32_fast = os.environ.get("CVVAE_F32_MODE", "exact") == "fast"  # see _CVVAEBase.fp32_mode

    def _cache(self) -> engine.WeightCache:
        if self._wc is None:
            object.__setattr__(self, "_wc", engine.WeightCache(self))
        self._wc.fast = self._fp32_fast
        return self._wc

    def _check_input(self, x: torch.Tensor):
        if x.dim() != 5:
            raise ValueError(f"expected a [B,C,T,H,W] tensor, got shape {tuple(x.shape)}")
        if self._unsupported:
            raise NotImplementedError(self._unsupported)
        if not x.is_cuda:
            raise RuntimeError("cvvae_amd runs on an MI355X (ROCm) device only; move the model and the input to 'cuda'. "
                               "There is no CPU fallback.")
        pdev = self.conv_in.weight.device
        if pdev != x.device:
            raise RuntimeError(f"input on {x.device} but the network's parameters are on {pdev}")

    _trainable = False  # networks with a backward pass on the kernels (grad3d.py): Encoder3D of the vae3d_sd3 family

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        self._check_input(x)
        # torch.autocast over an fp32 model (the reference's trainer with precision 16 / bf16: fp32 master weights, 16-bit compute;
        # main.py:905-912): the pass runs on 16-bit copies of the weights and returns the autocast dtype, as F.conv3d would
        # (the dtype is set for THIS pass and restored afterwards: WeightCache.computing_in)
        cd = self._pass_dtype(x)
        taped = (self._trainable and self.training and torch.is_grad_enabled()
                 and (x.requires_grad or any(p.requires_grad for p in self.parameters())))
        # parameters written through `.data` (EMA swaps: lvdm/modules/ema.py:61-86) do not move the cache's keys; a device-side
        # checksum does.  It costs a host sync (measured: the host then trails the GPU for the first ~7 ms of a training step), so the
        # TAPED pass runs it once after every train() / eval() transition (restore() precedes train()).  The inference branch -- where
        # the reference's validation_step / log_images run a plain pass and THEN enter ema_scope() in the same mode
        # (lvdm/models/autoencoder.py:379-384, 1193, 1426) -- runs it on EVERY pass while any parameter of the network still has
        # requires_grad, which is exactly the set LitEma.copy_to writes (ema.py:61-68); a frozen model (cvvae_inference_video.py:12:
        # `vae3d.requires_grad_(False)`) pays nothing.  `weight_guard` forces the per-pass check.
        if self.weight_guard or self._guard_pending or (not taped and self._ema_writable()):
            object.__setattr__(self, "_guard_pending", False)
            self.refresh_weights(only_if_changed=True)
        with self._cache().computing_in(cd):
            if taped:
                # training the codec itself (lvdm/models/autoencoder.py:1057-1090 runs the 3-D networks under autograd): the same
                # launches with a tape, two autograd nodes (body + tail) over (x, parameters).  eval() mode / no_grad: the inference pass below
                from . import grad3d
                return grad3d.run_trainable(self, x, kwargs)
            with torch.no_grad():
                return self._forward_inference(x, **kwargs)

    # fp32 models only: run THIS network's passes on 16-bit copies of its weights (torch.float16 / torch.bfloat16), as under
    # torch.autocast but without the context -- `CVVAEModel.decoder_compute_dtype` builds the mixed tolerance mode on it (fp32-fast
    # encoder: latents inside north_star's 1e-3 bound; 16-bit decoder).  None = the parameters' own dtype.
    compute_dtype_override: Optional[torch.dtype] = None
    # every pass re-checks a device-side checksum of the parameters first (one sync per pass) and drops stale packed forms: for
    # callers that write FROZEN weights through `.data` at arbitrary moments and cannot call refresh_weights() themselves.  Default:
    # after each train() / eval() transition, and on every inference-branch pass of a network with trainable parameters (see forward).
    weight_guard: bool = os.environ.get("CVVAE_WEIGHT_GUARD", "0") == "1"
    _guard_pending: bool = False

    def train(self, mode: bool = True):
        object.__setattr__(self, "_guard_pending", True)
        return super().train(mode)

    def _ema_writable(self) -> bool:
        """does any parameter still have requires_grad (what an EMA swap through `.data` can touch)?  Not during a hipGraph capture:
        the checksum synchronises"""
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return False
        return next((True for p in self.parameters() if p.requires_grad), False)

    def _pass_dtype(self, x: torch.Tensor) -> Optional[torch.dtype]:
        if self.conv_in.weight.dtype != torch.float32:
            return None
        ac = _autocast_dtype(x)
        return ac if ac is not None else self.compute_dtype_override

    def refresh_weights(self, only_if_changed: bool = False) -> bool:
        """Drop every packed / converted weight form and captured hipGraph of this network, so that the next pass re-reads the
        parameters.  Needed after writing parameters through `.data` (`p.data.copy_()` does not move `p._version`, which is what the
        cache is keyed on): the reference's EMA does exactly that (LitEma.copy_to / restore, lvdm/modules/ema.py:61-86).
        load_state_dict, optimizer steps, .to() and assignment are seen without it.  only_if_changed: decide by a device-side
        checksum of the parameters (one host sync).  Returns whether anything was dropped."""
        wc = self._cache()
        if only_if_changed:
            dropped = wc.guard()
        else:
            wc.invalidate()
            dropped = True
        if dropped and self._graphs is not None:
            self._graphs.clear()
        return dropped

    def _forward_inference(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        # every launch of the pass goes to x's device and its current stream, whatever the caller's current device is
        with torch.cuda.device(x.device):
            if self._graphs is not None and not torch.cuda.is_current_stream_capturing():
                return self._forward_graphed(x, **kwargs)
            return type(self)._program(self._cache(), x, dict(self._cfg, **kwargs))

    # ---- hipGraph replay of a whole encoder / decoder pass ------------------------------------------------------------
    # One pass is 40-110 kernel launches.  On clips the GPU time hides the host's launch work; on images and small tiles
    # (the T2I pipeline's decode(z, num_frames=1), BASELINE config 1) the pass is launch-bound.  With graphs enabled the
    # launch sequence of each input shape is captured once (the kernels never synchronise, never allocate and take every
    # argument by value, so the captured launches are exactly the eager ones) and replayed from then on: same kernels,
    # same instance choice, bit-identical results.  The reference has no counterpart (plain eager PyTorch).
    def enable_hip_graphs(self, enabled: bool = True, max_shapes: int = 4):
        """Capture-and-replay of this network's launch sequence per input shape (static input/output buffers and the pass's
        activations stay allocated per captured shape: keep `max_shapes` small; the oldest shape is dropped beyond it)."""
        object.__setattr__(self, "_graphs", {} if enabled else None)
        object.__setattr__(self, "_graph_cap", max(1, int(max_shapes)))
        return self

    def _forward_graphed(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        key = (tuple(x.shape), x.dtype, x.device.index, tuple(sorted(kwargs.items())), engine.switches_key(self._cache()), self._fp32_fast)
        ent = self._graphs.get(key)
        if ent is not None and ent[0] != sig:  # weights were replaced / moved / modified: the captured pointers are stale
            del self._graphs[key]
            ent = None
        if ent is None:
            while len(self._graphs) >= self._graph_cap:
                self._graphs.pop(next(iter(self._graphs)))
            prog, wc, cfg = type(self)._program, self._cache(), dict(self._cfg, **kwargs)
            static_in = x.detach().clone()
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):  # warm-up outside the capture: packs the weights, fills the device-query caches
                prog(wc, static_in, cfg)
            cur.wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = prog(wc, static_in, cfg)
            ent = (sig, graph, static_in, static_out)
            self._graphs[key] = ent
        _, graph, static_in, static_out = ent
        static_in.copy_(x)
        graph.replay()
        return static_out.clone()

    def get_last_layer(self, **kwargs):
        return self.conv_out.weight


# ------------------------------------------------------------------------------------------------------
# vae3d_sd3 graphs (vae_models3d_sd3.py:55-391)
# ------------------------------------------------------------------------------------------------------
def _sd3_mid(c: int, attention: bool) -> nn.Module:
    resnets = nn.ModuleList([_resnet(c, c, "conv_shortcut", (1, 1)), _resnet(c, c, "conv_shortcut", (1, 1))])
    atts = nn.ModuleList()
    if attention:
        a = _holder(group_norm=NormP(c), to_q=ConvP(c, c, ()), to_k=ConvP(c, c, ()), to_v=ConvP(c, c, ()))
        a.add_module("to_out", nn.ModuleList([ConvP(c, c, ()), nn.Identity()]))
        atts.append(a)
    return _holder(resnets=resnets, attentions=atts)


class Encoder3D(_Net):
    _program = staticmethod(engine.sd3_encoder)
    _trainable = True  # train() mode under grad mode: the taped pass + grad3d's body / tail backward as two autograd nodes

    def __init__(self, in_channels=3, out_channels=16, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 double_z=True, mid_block_add_attention=True, causal=True, **_):
        super().__init__()
        boc = list(block_out_channels)
        self.conv_in = ConvP(in_channels, boc[0], (3, 3, 3))
        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i, co in enumerate(boc):
            blk = _holder(resnets=nn.ModuleList(
                [_resnet(ch if j == 0 else co, co, "conv_shortcut", (1, 1)) for j in range(layers_per_block)]))
            ch = co
            if i != len(boc) - 1:
                blk.add_module("downsamplers", nn.ModuleList([_holder(conv=ConvP(co, co, (3, 3, 3)))]))
            self.down_blocks.append(blk)
        self.mid_block = _sd3_mid(boc[-1], mid_block_add_attention)
        self.conv_norm_out = NormP(boc[-1])
        self.conv_out = ConvP(boc[-1], 2 * out_channels if double_z else out_channels, (3, 3, 3))
        self._cfg = dict(causal=causal, block_out_channels=boc, layers_per_block=layers_per_block,
                         mid_block_add_attention=mid_block_add_attention)


class Decoder3D(_Net):
    _program = staticmethod(engine.sd3_decoder)
    _trainable = True

    def __init__(self, in_channels=16, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 mid_block_add_attention=True, causal=False, **_):
        super().__init__()
        boc = list(block_out_channels)
        rev = list(reversed(boc))
        self.conv_in = ConvP(in_channels, rev[0], (3, 3, 3))
        self.mid_block = _sd3_mid(rev[0], mid_block_add_attention)
        self.up_blocks = nn.ModuleList()
        ch = rev[0]
        for i, co in enumerate(rev):
            blk = _holder(resnets=nn.ModuleList(
                [_resnet(ch if j == 0 else co, co, "conv_shortcut", (1, 1)) for j in range(layers_per_block + 1)]))
            ch = co
            if i != len(rev) - 1:
                up_time = 2 if i % 2 == 0 else 1
                blk.add_module("upsamplers", nn.ModuleList([_holder(conv=ConvP(co, co * up_time, (3, 3, 3)))]))
            self.up_blocks.append(blk)
        self.conv_norm_out = NormP(boc[0])
        self.conv_out = ConvP(boc[0], out_channels, (3, 3, 3))
        self._cfg = dict(causal=causal, block_out_channels=boc, layers_per_block=layers_per_block,
                         mid_block_add_attention=mid_block_add_attention)


# ------------------------------------------------------------------------------------------------------
# vae3d graphs (vae_models.py:679-1002), LDM key naming
# ------------------------------------------------------------------------------------------------------
def _v3_attn(c: int, temporal: bool) -> nn.Module:
    a = _holder(norm=NormP(c), q=ConvP(c, c, (1, 1)), k=ConvP(c, c, (1, 1)), v=ConvP(c, c, (1, 1)),
                proj_out=ConvP(c, c, (1, 1)))
    if temporal:
        for n in ("q_t", "k_t", "v_t", "proj_out_t"):
            a.add_module(n, ConvP(c, c, ()))
        a.add_module("norm_t", NormP(c))
    return a


class Encoder(_Net):
    _program = staticmethod(engine.v3_encoder)
    _trainable = True  # (grad3d.py: the encoder of this family trains on the kernels; its decoder does not yet)

    def __init__(self, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4, double_z=True,
                 causal=True, **_):
        super().__init__()
        mult = list(ch_mult)
        in_mult = [1] + mult
        self.conv_in = ConvP(in_channels, ch, (3, 3, 3))
        self.down = nn.ModuleList()
        bi = ch
        for lvl in range(len(mult)):
            bi, bo = ch * in_mult[lvl], ch * mult[lvl]
            blocks = nn.ModuleList()
            for _j in range(num_res_blocks):
                blocks.append(_resnet(bi, bo, "nin_shortcut", (1, 1, 1)))
                bi = bo
            lv = _holder(block=blocks, attn=nn.ModuleList())
            if lvl != len(mult) - 1:
                lv.add_module("downsample", _holder(conv=ConvP(bi, bi, (3, 3, 3))))
            self.down.append(lv)
        self.mid = _holder(block_1=_resnet(bi, bi, "nin_shortcut", (1, 1, 1)), attn_1=_v3_attn(bi, False),
                           block_2=_resnet(bi, bi, "nin_shortcut", (1, 1, 1)))
        self.norm_out = NormP(bi)
        self.conv_out = ConvP(bi, 2 * z_channels if double_z else z_channels, (3, 3, 3))
        self._cfg = dict(causal=causal, ch_mult=mult, num_res_blocks=num_res_blocks)


class Decoder(_Net):
    _program = staticmethod(engine.v3_decoder)
    _trainable = True

    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, causal=False, **_):
        super().__init__()
        mult = list(ch_mult)
        bi = ch * mult[-1]
        self.conv_in = ConvP(z_channels, bi, (3, 3, 3))
        self.mid = _holder(block_1=_resnet(bi, bi, "nin_shortcut", (1, 1, 1)), attn_1=_v3_attn(bi, True),
                           block_2=_resnet(bi, bi, "nin_shortcut", (1, 1, 1)))
        ups = [None] * len(mult)
        for lvl in reversed(range(len(mult))):
            bo = ch * mult[lvl]
            blocks = nn.ModuleList()
            for _j in range(num_res_blocks + 1):
                blocks.append(_resnet(bi, bo, "nin_shortcut", (1, 1, 1)))
                bi = bo
            lv = _holder(block=blocks, attn=nn.ModuleList())
            if lvl != 0:
                up_time = 2 if lvl % 2 == 1 else 1
                lv.add_module("upsample", _holder(conv=ConvP(bi, bi * up_time, (3, 3, 3))))
            ups[lvl] = lv
        self.up = nn.ModuleList(ups)
        self.norm_out = NormP(bi)
        self.conv_out = ConvP(bi, out_ch, (3, 3, 3))
        self._cfg = dict(causal=causal, ch_mult=mult, num_res_blocks=num_res_blocks)
        self.last_z_shape = None

    def forward(self, z, **kwargs):
        self.last_z_shape = z.shape  # vae_models.py:962
        # (the reference ignores its kwargs; `u8_out` is this implementation's private switch of decode_to_frames_u8)
        return super().forward(z, **{k: v for k, v in kwargs.items() if k == "u8_out"})


# ------------------------------------------------------------------------------------------------------
# output types (diffusers' AutoencoderKLOutput / DecoderOutput / DiagonalGaussianDistribution contracts)
# ------------------------------------------------------------------------------------------------------
class _Output(dict):
    """diffusers' BaseOutput contract: attribute access, string keys, and integer / slice indexing over `to_tuple()`
    (`vae.decode(z)[0]`)."""

    def to_tuple(self):
        return tuple(self.values())

    def __getitem__(self, k):
        if isinstance(k, (int, slice)):
            return self.to_tuple()[k]
        return dict.__getitem__(self, k)


class AutoencoderKLOutput(_Output):
    def __init__(self, latent_dist):
        super().__init__(latent_dist=latent_dist)
        self.latent_dist = latent_dist


class DecoderOutput(_Output):
    def __init__(self, sample):
        super().__init__(sample=sample)
        self.sample = sample


class DiagonalGaussianDistribution:
    """Posterior over latents; maths of /root/reference/lvdm/modules/distributions/distributions.py:24-73 (the in-tree
    twin of diffusers' class) plus diffusers' `sample(generator=)` signature."""

    def __init__(self, parameters: torch.Tensor, deterministic: bool = False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        dev = self.parameters.device
        rand_dev = dev
        if generator is not None and generator.device.type == "cpu" and dev.type != "cpu":
            rand_dev = torch.device("cpu")  # diffusers randn_tensor: draw on the generator's device, then move
        noise = torch.randn(self.mean.shape, generator=generator, device=rand_dev, dtype=self.parameters.dtype).to(dev)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean

    def nll(self, sample: torch.Tensor, dims=(1, 2, 3)) -> torch.Tensor:
        if self.deterministic:
            return torch.Tensor([0.0])
        logtwopi = math.log(2.0 * math.pi)
        return 0.5 * torch.sum(logtwopi + self.logvar + torch.pow(sample - self.mean, 2) / self.var, dim=list(dims))

    def kl(self, other=None):
        dims = list(range(1, self.mean.dim()))
        if self.deterministic:
            return torch.Tensor([0.0])
        if other is None:
            return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=dims)
        return 0.5 * torch.sum(torch.pow(self.mean - other.mean, 2) / other.var + self.var / other.var - 1.0
                               - self.logvar + other.logvar, dim=dims)


# ------------------------------------------------------------------------------------------------------
# the wrapper (modeling_vae.py): config, from_pretrained, temporal windows, spatial tiles, blending
# ------------------------------------------------------------------------------------------------------
class _Config(SimpleNamespace):
    def __getitem__(self, k):
        return getattr(self, k)

    def get(self, k, default=None):
        return getattr(self, k, default)

    def to_dict(self):
        return dict(self.__dict__)


def _convert_deprecated_attention_keys(sd: dict) -> dict:
    """diffusers' ModelMixin.from_pretrained renames the parameters of attention blocks saved by its pre-0.18 `AttentionBlock`
    (`_convert_deprecated_attention_blocks`: query / key / value / proj_attn -> to_q / to_k / to_v / to_out.0) before the strict
    load; the sd3-family mid blocks (`Attention(..., _from_deprecated_attn_block=True)`, vae_blocks3d_sd3.py:806-822) are such
    blocks, so a checkpoint written by an older diffusers loads through the reference -- and through this loader -- unchanged."""
    ren = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}
    out = {}
    for k, v in sd.items():
        if ".attentions." in k:
            for old, new in ren.items():
                if old in k:
                    k = k.replace(old, new)
                    break
        out[k] = v
    return out


def _channel_constraints(widths) -> list:
    """Block widths the kernels can run: the fused GroupNorm statistics need a power-of-two number >= 4 of channels per group
    (32 groups) and the 1x1 convs (shortcuts, attention) consume 128-channel K chunks -> widths of 128 * 2^k."""
    bad = [int(w) for w in widths if w < 128 or w % 128 or ((w // 32) & (w // 32 - 1))]
    return [f"block widths must be 128 * 2^k (GroupNorm slots of the fused statistics, 128-channel K chunks of the 1x1 convs); got {bad}"] if bad else []


class _CVVAEBase(nn.Module):
    config_name = "config.json"
    _defaults: dict = {}
    _class_name = ""

    # ---- construction -------------------------------------------------------------------------------
    def _init_common(self, kw: dict):
        cfg = dict(self._defaults)
        unknown = [k for k in kw if k not in cfg and not k.startswith("_")]
        if unknown:
            raise TypeError(f"{type(self).__name__}: unexpected config keys {unknown}")
        cfg.update({k: v for k, v in kw.items() if not k.startswith("_")})
        self.config = _Config(**cfg)
        n, tnc = cfg["en_de_n_frames_a_time"], cfg["time_n_compress"]
        if n is not None:  # modeling_vae.py:84-91
            assert tnc is not None
            assert n % tnc == 0
            self.encode_n_frames_a_time, self.decode_n_frames_a_time = n, n // tnc
        else:
            self.encode_n_frames_a_time = self.decode_n_frames_a_time = None
        nvf = cfg["num_video_frames"]
        if nvf is not None:  # :93-99
            assert tnc is not None
            self.num_video_frames, self.num_latent_frames = nvf, 1 + (nvf - 1) // tnc
        else:
            self.num_video_frames = self.num_latent_frames = None
        ts = cfg["tile_spatial_size"]
        if ts is not None:  # :101-109
            assert cfg["spatial_n_compress"] is not None and cfg["tile_overlap_ratio"] is not None
            self.pixel_tile_size, self.latent_tile_size = ts, ts // cfg["spatial_n_compress"]
            self.tile_overlap_ratio = cfg["tile_overlap_ratio"]
        else:
            self.pixel_tile_size = self.latent_tile_size = self.tile_overlap_ratio = None
        self.reshape_z_dim_to_4 = cfg["reshape_z_dim_to_4"]  # stored, never applied in encode (Appendix A.1)
        self.reshape_x_dim_to_4 = cfg["reshape_x_dim_to_4"]

    # ---- arithmetic of fp32 models (torch_dtype=torch.float32, the reference's default).  gfx950 has no fp32-class MFMA, so an
    #      fp32 model runs in split precision on the 16-bit matrix pipe:
    #        "exact": three fp16 MFMAs per product -- ~1e-6 relative, latent max |delta| ~8e-6 against the reference's fp32 run
    #        "fast" : one fp16 MFMA + the two correction terms on the bf8 K = 64 MFMA -- 2/3 of the time of "exact", latent max
    #                 |delta| ~2e-4: the cheapest mode inside north_star's 1e-3 bound
    #      16-bit models ignore it.  Default "exact"; CVVAE_F32_MODE=fast changes the default.
    def _get_fp32_mode(self) -> str:
        return "fast" if self.encoder._fp32_fast else "exact"

    def _set_fp32_mode(self, mode: str):
        if mode not in ("exact", "fast"):
            raise ValueError(f"fp32_mode must be 'exact' or 'fast', got {mode!r}")
        for net in (self.encoder, self.decoder):
            object.__setattr__(net, "_fp32_fast", mode == "fast")

    fp32_mode = property(_get_fp32_mode, _set_fp32_mode)

    # ---- mixed tolerance mode (fp32 models): `north_star` bounds the LATENTS (|delta| <= 1e-3) and asks the frames to match "within fp16
    #      tolerance".  The latents are the encoder's output alone, so an fp32 model can keep its encoder in the fp32 arithmetic
    #      (`fp32_mode = "fast"`: latent max |delta| ~2e-4) and run the DECODER -- three quarters of the work -- on 16-bit copies of
    #      its weights with 16-bit activations (`decoder_compute_dtype = torch.float16`: reconstruction ~66 dB against the reference's
    #      fp32 frames, what the reference's own fp16 scripts deliver).  decode() then returns that dtype, as under torch.autocast.
    def _get_decoder_compute_dtype(self) -> Optional[torch.dtype]:
        return self.decoder.compute_dtype_override

    def _set_decoder_compute_dtype(self, dt: Optional[torch.dtype]):
        if dt not in (None, torch.float16, torch.bfloat16):
            raise ValueError(f"decoder_compute_dtype must be None, torch.float16 or torch.bfloat16, got {dt!r}")
        object.__setattr__(self.decoder, "compute_dtype_override", dt)

    decoder_compute_dtype = property(_get_decoder_compute_dtype, _set_decoder_compute_dtype)

    def refresh_weights(self, only_if_changed: bool = False) -> bool:
        """after writing parameters through `.data` (EMA swap, manual surgery): see _Net.refresh_weights"""
        a = self.encoder.refresh_weights(only_if_changed)
        b = self.decoder.refresh_weights(only_if_changed)
        return a or b

    def _flag_widths(self, widths):
        """widths the kernels cannot run: the model can still be built, loaded, converted and saved (parameter holder), but a
        forward pass raises NotImplementedError with this message instead of failing inside a launch."""
        bad = _channel_constraints(widths)
        if bad:
            warnings.warn(f"{type(self).__name__}: {bad[0]} -- this configuration can be loaded / saved but not run on the MI355X path")
            self.encoder._unsupported = self.decoder._unsupported = bad[0]

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder: Optional[str] = None,
                        torch_dtype: Optional[torch.dtype] = None, **kwargs):
        """diffusers ModelMixin.from_pretrained semantics for a local directory: <path>/<subfolder>/config.json +
        diffusion_pytorch_model.safetensors (or .bin), strict load, cast, eval  (cvvae_inference_video.py:11)."""
        root = os.fspath(pretrained_model_name_or_path)
        if subfolder:
            root = os.path.join(root, subfolder)
        cfg_path = os.path.join(root, cls.config_name)
        if not os.path.isfile(cfg_path):
            raise OSError(f"{cfg_path} not found (only local directories are supported: there is no hub access)")
        with open(cfg_path) as f:
            cfg = json.load(f)
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
        extra = sorted(k for k in cfg if k not in cls._defaults)
        if extra:  # diffusers' ConfigMixin ignores keys the class does not take
            warnings.warn(f"{cfg_path}: config keys {extra} are not used by {cls.__name__} and were ignored")
            cfg = {k: v for k, v in cfg.items() if k in cls._defaults}
        model = cls(**cfg)
        variant = kwargs.get("variant")  # diffusers: diffusion_pytorch_model.<variant>.safetensors (e.g. "fp16")
        stem = "diffusion_pytorch_model" + (f".{variant}" if variant else "")
        st = os.path.join(root, stem + ".safetensors")
        for idx in (st + ".index.json", os.path.join(root, stem + ".bin.index.json")):
            if os.path.isfile(idx):
                raise NotImplementedError(f"{idx}: sharded checkpoints are not supported; merge the shards into one "
                                          f"{stem}.safetensors (the CV-VAE checkpoints are single files of < 1 GB)")
        if os.path.isfile(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            binp = os.path.join(root, stem + ".bin")
            if not os.path.isfile(binp):
                raise OSError(f"no {stem}.safetensors/.bin under {root}")
            sd = torch.load(binp, map_location="cpu", weights_only=True)
        sd = _convert_deprecated_attention_keys(sd)
        own = model.state_dict()
        missing = sorted(k for k in own if k not in sd)
        bad_shape = sorted(k for k in own if k in sd and tuple(sd[k].shape) != tuple(own[k].shape))
        if missing or bad_shape:  # a half-initialised codec is never what the caller wants: fail with the full list
            raise RuntimeError(f"{root}: checkpoint does not fit {cls.__name__}: missing {missing[:8]}{'...' if len(missing) > 8 else ''}"
                               f", shape mismatch {bad_shape[:8]}")
        unexpected = sorted(k for k in sd if k not in own)
        if unexpected:  # diffusers warns about and drops keys the model does not own
            warnings.warn(f"{root}: {len(unexpected)} checkpoint tensors are not used by {cls.__name__}: {unexpected[:6]}...")
        model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=True)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        model.eval()
        return model

    def save_pretrained(self, save_directory, safe_serialization: bool = True):
        os.makedirs(save_directory, exist_ok=True)
        cfg = self.config.to_dict()
        cfg["_class_name"] = self._class_name or type(self).__name__
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "diffusion_pytorch_model.safetensors"))
        else:
            torch.save(sd, os.path.join(save_directory, "diffusion_pytorch_model.bin"))

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    # diffusers no-op toggles some pipelines call
    def enable_hip_graphs(self, enabled: bool = True, max_shapes: int = 4):
        """hipGraph capture-and-replay of the encoder / decoder passes per input shape (see _Net.enable_hip_graphs): for
        launch-bound inputs (images, small tiles).  Results are bit-identical to the eager launches."""
        self.encoder.enable_hip_graphs(enabled, max_shapes)
        self.decoder.enable_hip_graphs(enabled, max_shapes)
        return self

    def enable_slicing(self): pass
    def disable_slicing(self): pass
    def enable_tiling(self, *a, **k): pass
    def disable_tiling(self): pass

    # ---- blending (modeling_vae.py:321-341 / 647-667): in place on b, fp32 ramp ----------------------
    def blend_v(self, a: torch.Tensor, b: torch.Tensor, overlap_size: int) -> torch.Tensor:
        return self._blend(a, b, overlap_size, 0)

    def blend_h(self, a: torch.Tensor, b: torch.Tensor, overlap_size: int) -> torch.Tensor:
        return self._blend(a, b, overlap_size, 1)

    @staticmethod
    def _blend(a, b, o, axis):
        if not b.is_cuda:
            raise RuntimeError("cvvae_amd blends on the MI355X only (no CPU path)")
        o = min(a.shape[3 + axis], b.shape[3 + axis], o)  # modeling_vae.py:322, 333
        if o <= 0:
            return b
        if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
            # training through the tiled wrapper (Autoencoding3DEngine tiles under autograd, lvdm/models/autoencoder.py:809-974): the
            # blend is linear in both tiles -- same kernel forward, the ramp's adjoint backward
            from .grad3d import BlendFn
            return BlendFn.apply(a, b, o, axis)
        with torch.cuda.device(b.device):
            if a.is_contiguous() and b.is_contiguous():
                return ops.blend_(a, b, o, axis)
            ac, bc = a.contiguous(), b.contiguous()
            ops.blend_(ac, bc, o, axis)
            b.copy_(bc)
            return b

    # ---- spatial tiles (modeling_vae.py:144-191, 230-277) ---------------------------------------------
    @staticmethod
    def _tile_grid(H: int, W: int, tile: int, stride: int):
        """the reference's tile loop (modeling_vae.py:150-160): rows of (i, j, h, w) input windows, start step `stride`, a row /
        column loop ends at the first tile that reaches the edge"""
        rows = []
        for i in range(0, H, stride):
            cols = []
            for j in range(0, W, stride):
                cols.append((i, j, min(tile, H - i), min(tile, W - j)))
                if j + tile >= W:
                    break
            rows.append(cols)
            if i + tile >= H:
                break
        return rows

    def _assemble_tiles(self, rows, overlap_out, stride_out):
        """blend (in place, in the reference's order: every tile with its already blended upper, then left neighbour), crop
        all but the last tile of a row / column to `stride_out`, concatenate (modeling_vae.py:161-191).  rows: network outputs
        (NCDHW) in the order of _tile_grid."""
        res = []
        for i, cols in enumerate(rows):
            rc = []
            for j, t in enumerate(cols):
                # (the reference blends IN PLACE, so its `rows[i - 1][j]` / `row[j - 1]` are the already blended neighbours; naming
                #  them explicitly keeps that true for the out-of-place blend nodes of the autograd path)
                if i > 0:
                    t = self.blend_v(res[i - 1][j], t, overlap_out)
                if j > 0:
                    t = self.blend_h(rc[j - 1], t, overlap_out)
                rc.append(t)
            res.append(rc)
        out_rows = []
        for i, cols in enumerate(res):
            for j, t in enumerate(cols):
                if i < len(res) - 1:
                    t = t[:, :, :, :stride_out, :]
                if j < len(cols) - 1:
                    t = t[:, :, :, :, :stride_out]
                cols[j] = t
            out_rows.append(torch.cat(cols, dim=4))
        return torch.cat(out_rows, dim=3)

    def _spatial_tiled(self, x, net, tile, stride, overlap_out, stride_out, hdim=3, wdim=4, **kwargs):
        """hdim / wdim: the H and W axes of the INPUT (3, 4 for NCDHW; 2, 3 for the NDHWC clips of encode_frames_u8); the
        network outputs are always NCDHW."""
        grid = self._tile_grid(x.shape[hdim], x.shape[wdim], tile, stride)
        rows = [[net(x.narrow(hdim, i, h).narrow(wdim, j, w)) for (i, j, h, w) in cols] for cols in grid]
        return self._assemble_tiles(rows, overlap_out, stride_out)

    def _tile_params(self, encode: bool):
        """(input tile, input stride, output overlap, output stride) of the spatial tiling, or None when tiling is off"""
        if self.pixel_tile_size is None:
            return None
        if encode:
            ov = round(self.latent_tile_size * self.tile_overlap_ratio)
            return (self.pixel_tile_size, round(self.pixel_tile_size * (1 - self.tile_overlap_ratio)), ov, self.latent_tile_size - ov)
        ov = round(self.pixel_tile_size * self.tile_overlap_ratio)
        return (self.latent_tile_size, round(self.latent_tile_size * (1 - self.tile_overlap_ratio)), ov, self.pixel_tile_size - ov)

    def spatial_tiled_encode(self, x, _ndhwc=False):
        # _ndhwc (private): x is a channel-padded NDHWC clip [B,T,H,W,Cpad] straight from the device-side pre-processing
        net = (lambda t: self.encoder(t.contiguous(), ndhwc_in=True)) if _ndhwc else self.encoder
        if self.pixel_tile_size is None:
            return net(x)
        pixel_stride = round(self.pixel_tile_size * (1 - self.tile_overlap_ratio))
        latent_overlap = round(self.latent_tile_size * self.tile_overlap_ratio)
        return self._spatial_tiled(x, net, self.pixel_tile_size, pixel_stride, latent_overlap,
                                   self.latent_tile_size - latent_overlap, *((2, 3) if _ndhwc else (3, 4)))

    def spatial_tiled_decode(self, z, **kwargs):
        if self.latent_tile_size is None:
            return self.decoder(z, **kwargs)
        latent_stride = round(self.latent_tile_size * (1 - self.tile_overlap_ratio))
        pixel_overlap = round(self.pixel_tile_size * self.tile_overlap_ratio)
        return self._spatial_tiled(z, self.decoder, self.latent_tile_size, latent_stride, pixel_overlap,
                                   self.pixel_tile_size - pixel_overlap)

    # ---- temporal windows (modeling_vae.py:193-210, 279-296) -------------------------------------------
    @staticmethod
    def _windows(T: int, stride: int):
        n_rounds = math.ceil((T - 1) / stride)
        n_rounds = 1 if n_rounds == 0 else n_rounds
        return [(n * stride, (n + 1) * stride + 1) for n in range(n_rounds)]

    def tiled_encode(self, x, _ndhwc=False):
        if self.encode_n_frames_a_time is None:
            return self.spatial_tiled_encode(x, _ndhwc)
        assert x.dim() == 5
        outs = []
        tdim = 1 if _ndhwc else 2
        for n, (a, b) in enumerate(self._windows(x.shape[tdim], self.encode_n_frames_a_time)):
            z_i = self.spatial_tiled_encode(x.narrow(tdim, a, min(b, x.shape[tdim]) - a), _ndhwc)
            outs.append(z_i if n == 0 else z_i[:, :, 1:])
        return torch.cat(outs, dim=2)

    def tiled_decode(self, z, **kwargs):
        if self.decode_n_frames_a_time is None:
            return self.spatial_tiled_decode(z, **kwargs)
        assert z.dim() == 5
        outs = []
        for n, (a, b) in enumerate(self._windows(z.shape[2], self.decode_n_frames_a_time)):
            x_i = self.spatial_tiled_decode(z[:, :, a:b], **kwargs)
            outs.append(x_i if n == 0 else x_i[:, :, 1:])
        return torch.cat(outs, dim=2)

    # ---- public API (modeling_vae.py:212-228, 298-319, 114-142) ----------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        if x.dim() == 4:
            if self.num_video_frames is not None:
                b = x.shape[0] // self.num_video_frames
                x = x.reshape(b, self.num_video_frames, *x.shape[1:]).permute(0, 2, 1, 3, 4)
            else:
                x = x.unsqueeze(2)
        moments = self.tiled_encode(x)
        posterior = DiagonalGaussianDistribution(moments)
        if not return_dict:
            return (posterior,)
        return AutoencoderKLOutput(latent_dist=posterior)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, num_frames: Optional[int] = None, return_dict: bool = True):
        if z.dim() == 4:
            t = num_frames if num_frames is not None else self.num_latent_frames
            if t is not None:
                z = z.reshape(z.shape[0] // t, t, *z.shape[1:]).permute(0, 2, 1, 3, 4)
            else:
                z = z.unsqueeze(2)
        x = self.tiled_decode(z)
        if self.reshape_x_dim_to_4:
            x = x.permute(0, 2, 1, 3, 4).reshape(-1, x.shape[1], *x.shape[3:])
        if not return_dict:
            return (x,)
        return DecoderOutput(sample=x)

    # ---- persistent packed-weight cache ("weight pre-packing cache") ----------------------------------
    def save_packed_weights(self, path: str) -> int:
        """Write the packed (MFMA fragment order) weight forms both networks have built so far -- i.e. after a pass over the shapes
        of interest -- to `path` (torch.save).  The packed format is private to the library version (ABI number stored)."""
        from . import _lib as L
        blob = {"abi": L.ABI_VERSION, "class": type(self).__name__,
                "nets": {n: getattr(self, n)._cache().export_packed() for n in ("encoder", "decoder")}}
        torch.save(blob, path)
        return sum(len(v) for v in blob["nets"].values())

    def load_packed_weights(self, path: str) -> int:
        """Install packed weights written by `save_packed_weights` for every layer whose parameters still carry the recorded
        fingerprint (shape, dtype, three moments) -- e.g. after `from_pretrained` of the same checkpoint and `.to(dtype).cuda()`;
        anything else (another checkpoint, another library version) is packed on demand as usual.  Returns the number installed."""
        from . import _lib as L
        blob = torch.load(path, map_location="cpu", weights_only=True)
        if blob.get("abi") != L.ABI_VERSION or blob.get("class") != type(self).__name__:
            warnings.warn(f"{path}: packed weights of {blob.get('class')} / ABI {blob.get('abi')}, this is {type(self).__name__} / ABI "
                          f"{L.ABI_VERSION}: ignored")
            return 0
        return sum(getattr(self, n)._cache().import_packed(blob["nets"].get(n, {})) for n in ("encoder", "decoder"))

    # ---- device-side pixel pre/post-processing of the inference scripts ------------------------
    @torch.no_grad()
    def encode_frames_u8(self, frames: torch.Tensor, return_dict: bool = True, size: Optional[Tuple[int, int]] = None):
        """frames: uint8 [T,H,W,3] on the device (decord's layout).  Equivalent to the scripts' host-side
        `rearrange -> .half() -> / 127.5 - 1.0 -> [:, :, :frame_end]` (cvvae_inference_video.py:24-38) followed by
        `encode`; the normalisation runs on the MI355X in the model's dtype with the scripts' rounding steps."""
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise ValueError(f"expected uint8 frames [T,H,W,3], got {frames.dtype} {tuple(frames.shape)}")
        if size is not None:  # the scripts' transforms.Resize(size=(height, width)) on the uint8 clip (:14-16, 28), on the device
            with torch.cuda.device(frames.device):
                frames = ops.resize_frames_u8(frames.contiguous(), size)
        T = frames.shape[0]
        frame_end = 1 + (T - 1) // 4 * 4
        # the NDHWC clip, channel-padded for conv_in's K chunk (32 on the single-frame fold path), goes to the encoder as it is:
        # the reference's NCDHW clip is never built (windows / tiles are cut as views of it)
        with torch.cuda.device(frames.device):
            x = ops.frames_u8_to_ndhwc(frames[:frame_end].contiguous(), 32 if frame_end == 1 else 16, self.dtype)
            posterior = DiagonalGaussianDistribution(self.tiled_encode(x, _ndhwc=True))
        if not return_dict:
            return (posterior,)
        return AutoencoderKLOutput(latent_dist=posterior)

    @torch.no_grad()
    def encode_latents(self, x: torch.Tensor, sample: bool = True, generator: Optional[torch.Generator] = None,
                       scale_factor: Optional[float] = None, n_samples_a_time: Optional[int] = None,
                       flatten_frames: bool = False) -> torch.Tensor:
        """Frozen-encoder latent pre-compute for the diffusion training engines: the arithmetic of
        `DiffusionEngine.encode_first_stage` (/root/reference/lvdm/models/diffusion.py:159-171: rounds of
        `en_and_decode_n_samples_a_time` samples, `first_stage_model.encode`, cat, `scale_factor * z`) with the 4-D <-> 5-D
        adapters of `DiffusionEngineFor3DVAE.encode_first_stage` (:380-385: images [B,C,H,W] run as one-frame clips and the
        latents come back as [(B T),C,h,w]).  `sample=False` takes the posterior mode (deterministic latents for a cache).
        x: [B,3,T,H,W] clips or [B,3,H,W] images -> latents [B,z,T',h,w] (clips) / [(B T'),z,h,w] (images).
        DEVIATION, stated: the reference's :380-385 rearranges 'b c t h w -> (b t) c h w' for clips too; here clips come back 5-D
        (what a latent cache stores) unless `flatten_frames=True`, which gives the reference's [(B T'),z,h,w] for every input."""
        images = x.dim() == 4
        if images:
            x = x.unsqueeze(2)
        n = x.shape[0] if n_samples_a_time is None else int(n_samples_a_time)
        outs = []
        for a in range(0, x.shape[0], n):
            post = DiagonalGaussianDistribution(self.tiled_encode(x[a:a + n]))
            outs.append(post.sample(generator=generator) if sample else post.mode())
        z = torch.cat(outs, dim=0)
        sf = scale_factor if scale_factor is not None else getattr(self.config, "scaling_factor", None)
        if sf is not None and sf != 1.0:
            z = sf * z
        if images or flatten_frames:
            z = z.permute(0, 2, 1, 3, 4).reshape(-1, z.shape[1], *z.shape[3:])
        return z

    @torch.no_grad()
    def decode_latents(self, z: torch.Tensor, scale_factor: Optional[float] = None, n_samples_a_time: Optional[int] = None,
                       flatten_frames: bool = True) -> torch.Tensor:
        """The other half of the training / sampling engines' first-stage contract: `DiffusionEngine.decode_first_stage`
        (/root/reference/lvdm/models/diffusion.py:139-157: `z / scale_factor`, rounds of `en_and_decode_n_samples_a_time` samples,
        `first_stage_model.decode`, cat) with the adapters of `DiffusionEngineFor3DVAE.decode_first_stage` (:369-377: image latents
        [B,z,h,w] run as one-frame clips; the frames come back as [(B T),3,H,W]).  `flatten_frames=False` keeps clips 5-D."""
        if z.dim() == 4:
            z = z.unsqueeze(2)
        sf = scale_factor if scale_factor is not None else getattr(self.config, "scaling_factor", None)
        if sf is not None and sf != 1.0:
            z = (1.0 / sf) * z
        n = z.shape[0] if n_samples_a_time is None else int(n_samples_a_time)
        x = torch.cat([self.decode(z[a:a + n]).sample for a in range(0, z.shape[0], n)], dim=0)
        if flatten_frames and x.dim() == 5:
            x = x.permute(0, 2, 1, 3, 4).reshape(-1, x.shape[1], *x.shape[3:])
        return x

    @torch.no_grad()
    def decode_to_frames_u8(self, z: torch.Tensor, num_frames: Optional[int] = None) -> torch.Tensor:
        """`decode(z).sample` followed by the scripts' `(clamp(x,-1,1)+1)*127.5 -> uint8` ('t h w c',
        cvvae_inference_video.py:47-50) on the device.  One clip (B = 1)."""
        # one window, one tile (clips up to 17 frames and 576 x 576 pixels): the uint8 conversion is the store of the decoder's last
        # pass (engine.decoder_conv_out, u8) -- the float clip is never written
        if (z.dim() == 5 and z.shape[0] == 1 and not self.reshape_x_dim_to_4 and
                (self.decode_n_frames_a_time is None or len(self._windows(z.shape[2], self.decode_n_frames_a_time)) == 1) and
                (self.latent_tile_size is None or (z.shape[3] <= self.latent_tile_size and z.shape[4] <= self.latent_tile_size))):
            out = self.decoder(z, u8_out=True)
            if out.dtype == torch.uint8:
                return out
            x = out
        else:
            x = self.decode(z, num_frames=num_frames).sample
        if x.dim() != 5 or x.shape[0] != 1:
            raise ValueError("decode_to_frames_u8 handles one clip [1,C,T,H,W] (reshape_x_dim_to_4 must be off)")
        with torch.cuda.device(x.device):  # (the launch goes to x's device whatever the caller's current device is)
            return ops.ncdhw_to_frames_u8(x.contiguous())

    def forward(self, sample: torch.Tensor, sample_posterior: bool = False, return_dict: bool = True,
                generator: Optional[torch.Generator] = None, num_frames: Optional[int] = None
                ) -> Union[DecoderOutput, Tuple[torch.Tensor]]:
        posterior = self.encode(sample).latent_dist
        z = posterior.sample(generator=generator) if sample_posterior else posterior.mode()
        dec = self.decode(z, num_frames=num_frames).sample
        if not return_dict:
            return (dec,)
        return DecoderOutput(sample=dec)


class CVVAEModel(_CVVAEBase):
    """'vae3d' (SD2.1-compatible, 4-ch latent): /root/reference/models/modeling_vae.py:20-341."""

    _class_name = "CVVAEModel"
    _defaults = dict(
        double_z=True, z_channels=4, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
        attn_resolutions=[], dropout=0.0, use_3d_conv=True, half_3d=True, causal_encoder=True, causal_decoder=False,
        encoder_attn_type="vanilla-xformers", decoder_attn_type="spatial-temporal-xformer", scaling_factor=0.18215,
        force_upcast=True, en_de_n_frames_a_time=16, time_n_compress=4, spatial_n_compress=8, tile_spatial_size=576,
        num_video_frames=None, tile_overlap_ratio=0.2222, reshape_z_dim_to_4=False, reshape_x_dim_to_4=False)

    def __init__(self, **kwargs):
        super().__init__()
        self._init_common(kwargs)
        c = self.config
        unsupported = []
        if not c.use_3d_conv or not c.half_3d:
            unsupported.append("use_3d_conv/half_3d must be True")
        if list(c.attn_resolutions):
            unsupported.append("attn_resolutions must be []")
        if c.dropout != 0.0:
            unsupported.append("dropout must be 0")
        if c.encoder_attn_type not in ("vanilla-xformers", "vanilla") or c.decoder_attn_type != "spatial-temporal-xformer":
            unsupported.append("attention types other than the shipped vanilla(-xformers)/spatial-temporal-xformer")
        if unsupported:
            raise NotImplementedError("CVVAEModel on MI355X supports the shipped CV-VAE configuration only: " + "; ".join(unsupported))
        self.encoder = Encoder(ch=c.ch, ch_mult=c.ch_mult, num_res_blocks=c.num_res_blocks, in_channels=c.in_channels,
                               z_channels=c.z_channels, double_z=c.double_z, causal=c.causal_encoder)
        self.decoder = Decoder(ch=c.ch, out_ch=c.out_ch, ch_mult=c.ch_mult, num_res_blocks=c.num_res_blocks,
                               z_channels=c.z_channels, causal=c.causal_decoder)
        self._flag_widths([c.ch * m for m in c.ch_mult])


class CVVAESD3Model(_CVVAEBase):
    """'vae3d_sd3' (SD3-compatible, 16-ch latent): /root/reference/models/modeling_vae.py:344-667."""

    _class_name = "CVVAESD3Model"
    _defaults = dict(
        in_channels=3, out_channels=16, down_block_types=["DownEncoderBlock3D"] * 4, up_block_types=["UpDecoderBlock3D"] * 4,
        block_out_channels=[128, 256, 512, 512], layers_per_block=2, norm_num_groups=32, act_fn="silu", double_z=True,
        mid_block_add_attention=True, causal_encoder=True, causal_decoder=False, half_3d=True, en_de_n_frames_a_time=16,
        time_n_compress=4, spatial_n_compress=8, tile_spatial_size=576, num_video_frames=None, tile_overlap_ratio=0.2222,
        reshape_z_dim_to_4=False, reshape_x_dim_to_4=False)

    def __init__(self, **kwargs):
        super().__init__()
        self._init_common(kwargs)
        c = self.config
        unsupported = []
        if any(t != "DownEncoderBlock3D" for t in c.down_block_types) or any(t != "UpDecoderBlock3D" for t in c.up_block_types):
            unsupported.append("block types other than DownEncoderBlock3D/UpDecoderBlock3D")
        if c.norm_num_groups != 32 or c.act_fn not in ("silu", "swish") or not c.half_3d:
            unsupported.append("norm_num_groups != 32, act_fn != silu or half_3d=False")
        if unsupported:
            raise NotImplementedError("CVVAESD3Model on MI355X supports the shipped CV-VAE configuration only: " + "; ".join(unsupported))
        self.encoder = Encoder3D(in_channels=c.in_channels, out_channels=c.out_channels, block_out_channels=c.block_out_channels,
                                 layers_per_block=c.layers_per_block, double_z=c.double_z,
                                 mid_block_add_attention=c.mid_block_add_attention, causal=c.causal_encoder)
        self.decoder = Decoder3D(in_channels=c.out_channels, out_channels=c.in_channels, block_out_channels=c.block_out_channels,
                                 layers_per_block=c.layers_per_block, mid_block_add_attention=c.mid_block_add_attention,
                                 causal=c.causal_decoder)
        self._flag_widths(c.block_out_channels)


# `north_star` calls the class AutoencoderKLCVVAE; the reference has no such name -- provide the alias.
AutoencoderKLCVVAE = CVVAESD3Model
