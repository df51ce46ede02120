"""Multi-GPU execution of the codec: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests).

The reference's wrapper already cuts a clip into INDEPENDENT 17-frame windows (stride 16, 1 shared frame;
/root/reference/models/modeling_vae.py:193-210, 279-296): GroupNorm statistics and the causal padding never cross a
window, so sharding windows across ranks is reference-exact.  Two entry styles:

  * replicated input  -- every rank holds the clip, codes its own contiguous run of windows, `all_gather`s the result;
  * time-sharded input -- rank r holds only its own frames; the single boundary frame each window run shares with its
    left neighbour travels by point-to-point send/recv ("causal halo exchange": 5.5 MB of pixels for encode at
    720x1280, 0.46 MB of latents for decode) -- no ring, no all-reduce.

There is no collective inside the network itself.
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def split_contiguous(n: int, world: int, rank: int) -> Tuple[int, int]:
    """balanced contiguous range [a, b) of n units for `rank` (first n % world ranks get one extra)."""
    q, r = divmod(n, world)
    a = rank * q + min(rank, r)
    return a, a + q + (1 if rank < r else 0)


def n_windows(T: int, stride: int) -> int:
    n = -(-(T - 1) // stride)
    return 1 if n == 0 else n


def owned_frames(T: int, stride: int, world: int, rank: int) -> Tuple[int, int]:
    """frame range [a, b) of a T-frame clip that `rank` owns under time sharding: its windows' frames minus the
    boundary frame that belongs to the left neighbour."""
    w0, w1 = split_contiguous(n_windows(T, stride), world, rank)
    if w0 == w1:
        return 0, 0
    a = w0 * stride + (0 if w0 == 0 else 1)
    b = min(w1 * stride + 1, T)
    return a, b


def _gather_time(local: torch.Tensor, counts: List[int], group=None) -> torch.Tensor:
    """all_gather along dim 2 of per-rank tensors whose dim-2 sizes are `counts` (known to every rank)."""
    world = dist.get_world_size(group)
    mx = max(counts)
    shp = list(local.shape)
    shp[2] = mx
    pad = local.new_zeros(shp)
    if local.shape[2]:
        pad[:, :, :local.shape[2]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([b[:, :, :c] for b, c in zip(bufs, counts) if c], dim=2)


def _exchange_boundary(x_local: torch.Tensor, have: List[bool], group=None) -> Optional[torch.Tensor]:
    """every rank with frames sends its LAST frame to the next rank that has frames; returns the received frame."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    active = [r for r in range(world) if have[r]]
    if rank not in active:
        return None
    i = active.index(rank)
    reqs, recv = [], None
    if i + 1 < len(active):
        reqs.append(dist.isend(x_local[:, :, -1:].contiguous(), dst=active[i + 1], group=group))
    if i > 0:
        recv = torch.empty_like(x_local[:, :, :1]).contiguous()
        reqs.append(dist.irecv(recv, src=active[i - 1], group=group))
    for q in reqs:
        q.wait()
    return recv


def _run_windows(fn, x: torch.Tensor, stride: int, first_is_global_first: bool) -> torch.Tensor:
    """code a run of consecutive windows held in x (x starts at a window boundary frame); drop output frame 0 of every
    window except the clip's very first one (modeling_vae.py:206, 293)."""
    outs = []
    for n in range(n_windows(x.shape[2], stride)):
        o = fn(x[:, :, n * stride:(n + 1) * stride + 1])
        outs.append(o if (n == 0 and first_is_global_first) else o[:, :, 1:])
    return torch.cat(outs, dim=2)


def _sharded(model, x: torch.Tensor, T_total: int, encode: bool, time_sharded: bool, gather: bool, group=None):
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    stride = model.encode_n_frames_a_time if encode else model.decode_n_frames_a_time
    fn = model.spatial_tiled_encode if encode else model.spatial_tiled_decode
    if stride is None:
        raise ValueError("window sharding needs en_de_n_frames_a_time (the reference's temporal chunking)")
    ranges = [owned_frames(T_total, stride, world, r) for r in range(world)]
    a, b = ranges[rank]
    if time_sharded:
        assert x.shape[2] == b - a, f"rank {rank} must hold frames [{a},{b}) of the clip"
        halo = _exchange_boundary(x, [rb > ra for ra, rb in ranges], group)
        mine = x if halo is None else torch.cat([halo, x], dim=2)
    else:
        assert x.shape[2] == T_total
        mine = x[:, :, max(a - 1, 0):b] if b > a else x[:, :, :0]
    if b > a:
        out = _run_windows(fn, mine, stride, first_is_global_first=(a == 0))
    else:
        out = None
    if not gather:
        return out
    # output frames per rank: encode -> latent frames, decode -> pixel frames
    f = (lambda n: 1 + (n - 1) // model.config.time_n_compress) if encode else (lambda n: 1 + (n - 1) * model.config.time_n_compress)
    counts = []
    for ra, rb in ranges:
        if rb <= ra:
            counts.append(0)
        else:
            span = rb - ra + (1 if ra > 0 else 0)          # frames seen incl. the halo
            counts.append(f(span) - (1 if ra > 0 else 0))  # minus the dropped first output frame
    if out is None:
        ref_shape = [0] * 5
        shp = torch.tensor(ref_shape, device=x.device)
    else:
        shp = torch.tensor(list(out.shape), device=x.device)
    shapes = [torch.empty_like(shp) for _ in range(world)]
    dist.all_gather(shapes, shp, group=group)
    proto = next(s for s in shapes if int(s[2]) > 0).tolist()
    if out is None:
        out = x.new_zeros((proto[0], proto[1], 0, proto[3], proto[4]))
    return _gather_time(out, counts, group)


def encode_windows_sharded(model, x: torch.Tensor, T_total: Optional[int] = None, time_sharded: bool = False,
                           gather: bool = True, group=None):
    """moments of the whole clip (every rank) or of this rank's windows (gather=False).
    x: the full clip (replicated) or this rank's `owned_frames` (time_sharded=True, then pass T_total)."""
    return _sharded(model, x, x.shape[2] if T_total is None else T_total, True, time_sharded, gather, group)


def decode_windows_sharded(model, z: torch.Tensor, T_total: Optional[int] = None, time_sharded: bool = False,
                           gather: bool = True, group=None):
    return _sharded(model, z, z.shape[2] if T_total is None else T_total, False, time_sharded, gather, group)
