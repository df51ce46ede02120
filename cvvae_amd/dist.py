"""Multi-GPU execution of the codec: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests).

The reference's wrapper already cuts a clip into INDEPENDENT 17-frame windows (stride 16, 1 shared frame;
/root/reference/models/modeling_vae.py:193-210, 279-296): GroupNorm statistics and the causal padding never cross a
window, so sharding windows across ranks is reference-exact.  Two entry styles:

  * replicated input  -- every rank holds the clip, codes its own contiguous run of windows, `all_gather`s the result;
  * time-sharded input -- rank r holds only its own frames; the single boundary frame each window run shares with its
    left neighbour travels by point-to-point send/recv ("causal halo exchange": 5.5 MB of pixels for encode at
    720x1280, 0.46 MB of latents for decode) -- no ring, no all-reduce.

  * window x tile units (`encode_units_sharded` / `decode_units_sharded`, replicated input) -- the finer partition SURVEY 8(e)
    names: every (window, spatial tile) network call is one unit (cfg 4: 8 x 6 = 48), split over the ranks in contiguous,
    area-balanced runs; raw tile results travel point-to-point to the rank that owns the window, which blends and crops them
    in the reference's order (modeling_vae.py:161-191).  A 17-frame 720p clip (ONE window) then keeps 6 GPUs busy, a 65-frame
    one all 8.

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
    TRAFFIC["sent"] += _nbytes(pad) * (world - 1)
    TRAFFIC["recv"] += _nbytes(pad) * (world - 1)
    return torch.cat([b[:, :, :c] for b, c in zip(bufs, counts) if c], dim=2)


def _peer(group, r: int) -> int:
    """P2POp / isend / irecv take GLOBAL ranks whatever `group` is: translate a group-relative rank."""
    return r if group is None else dist.get_global_rank(group, r)


# bytes this process has put on / taken off the wire through the point-to-point and gather calls of this module (diagnostic:
# bench.py prints them per step; reset with TRAFFIC.update(sent=0, recv=0))
TRAFFIC = {"sent": 0, "recv": 0}


def _nbytes(t: torch.Tensor) -> int:
    return t.numel() * t.element_size()


def _exchange_boundary(x_local: torch.Tensor, have: List[bool], group=None) -> Optional[torch.Tensor]:
    """every rank with frames sends its LAST frame to the next rank that has frames; returns the received frame.
    Send and receive are posted as ONE batch (`batch_isend_irecv`: on RCCL a single group call, so the N-1 transfers of a
    clip run concurrently instead of serialising into a chain of blocking pairs)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    active = [r for r in range(world) if have[r]]
    if rank not in active:
        return None
    i = active.index(rank)
    p2p, recv = [], None
    if i + 1 < len(active):
        last = x_local[:, :, -1:].contiguous()
        p2p.append(dist.P2POp(dist.isend, last, _peer(group, active[i + 1]), group))
        TRAFFIC["sent"] += _nbytes(last)
    if i > 0:
        recv = torch.empty_like(x_local[:, :, :1]).contiguous()
        p2p.append(dist.P2POp(dist.irecv, recv, _peer(group, active[i - 1]), group))
        TRAFFIC["recv"] += _nbytes(recv)
    if p2p:
        for q in dist.batch_isend_irecv(p2p):
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


@torch.no_grad()  # (inference entry points: a module left in train() mode must not take the taped training path here)
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
    if any(rb <= ra for ra, rb in ranges):
        # a rank without a window has to learn the result's geometry from one that has (a host sync; only when the clip has fewer
        # windows than there are ranks -- otherwise nothing here waits for the device)
        shp = torch.tensor([0] * 5 if out is None else list(out.shape), device=x.device)
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


@torch.no_grad()  # (inference entry points: a module left in train() mode must not take the taped training path here)
def codec_step_time_sharded(model, x_local: torch.Tensor, T_total: int, group=None):
    """encode + decode of ONE clip whose frames arrive sharded on time (rank r holds `owned_frames(T_total, 16, world, r)`):
    the step `bench.py --gpus N` times and the gloo / RCCL tests check.

      1. causal halo exchange: the last pixel frame of every rank goes to its right neighbour (one batched send/recv);
      2. every rank encodes its own windows (modeling_vae.py:519-536) -- no collective inside the network;
      3. the posterior moments are all-gathered on time (the clip's latent: what a caller of `encode` gets back; 15 MB at cfg 4);
      4. every rank decodes its own run of the 5-latent-frame windows (modeling_vae.py:605-622) out of the gathered latent -- the
         boundary latent frame is already there, nothing else travels; the pixels stay sharded on time.

    -> (moments of the whole clip [every rank], this rank's reconstructed frames or None).  Concatenated over the ranks the
    frames equal `model.decode(model.encode(x).latent_dist.mode()).sample` bit for bit."""
    mom = encode_windows_sharded(model, x_local, T_total=T_total, time_sharded=True, gather=True, group=group)
    z = mom[:, :mom.shape[1] // 2]
    y_local = decode_windows_sharded(model, z, time_sharded=False, gather=False, group=group)
    return mom, y_local


def decoded_frames_of_rank(model, T_latent: int, world: int, rank: int) -> Tuple[int, int]:
    """pixel-frame range [a, b) of the clip that `rank` holds after `codec_step_time_sharded`"""
    a, b = owned_frames(T_latent, model.decode_n_frames_a_time, world, rank)
    if b <= a:
        return 0, 0
    tnc = model.config.time_n_compress
    return (0 if a == 0 else (a - 1) * tnc + 1), (b - 1) * tnc + 1


# ------------------------------------------------------------------------------------------------------------------------
# window x tile units
# ------------------------------------------------------------------------------------------------------------------------
def balanced_runs(cost: List[int], world: int) -> List[Tuple[int, int]]:
    """contiguous runs [a, b) of the units, one per rank, with the smallest possible maximum run cost (linear partition: binary
    search on the bound, greedy fill); ranks left over are given work by splitting the costliest multi-unit runs, so that as
    many ranks as there are units are busy.  A rank may get nothing when there are fewer units than ranks."""
    n = len(cost)

    def fill(limit):
        runs, a, acc = [], 0, 0
        for i, c in enumerate(cost):
            if acc and acc + c > limit:
                runs.append((a, i))
                a, acc = i, 0
            acc += c
        runs.append((a, n))
        return runs

    lo, hi = max(cost), sum(cost)
    while lo < hi:
        mid = (lo + hi) // 2
        if len(fill(mid)) <= world:
            hi = mid
        else:
            lo = mid + 1
    runs = fill(lo)
    run_cost = lambda r: sum(cost[r[0]:r[1]])  # noqa: E731
    while len(runs) < world:
        multi = [r for r in runs if r[1] - r[0] > 1]
        if not multi:
            break
        a, b = max(multi, key=run_cost)
        # split where the two halves are closest
        best = min(range(a + 1, b), key=lambda k: abs(sum(cost[a:k]) - sum(cost[k:b])))
        i = runs.index((a, b))
        runs[i:i + 1] = [(a, best), (best, b)]
    runs += [(n, n)] * (world - len(runs))
    return runs


def unit_plan(model, shape, encode: bool, world: int):
    """-> (windows [(a, b)], tile grid rows [(i, j, h, w)], units [(n, r, c)], owner of every unit, owner of every window)"""
    T, H, W = shape[2], shape[3], shape[4]
    tstride = model.encode_n_frames_a_time if encode else model.decode_n_frames_a_time
    wins = [(a, min(b, T)) for a, b in model._windows(T, tstride)] if tstride is not None else [(0, T)]
    tp = model._tile_params(encode)
    grid = model._tile_grid(H, W, tp[0], tp[1]) if tp is not None else [[(0, 0, H, W)]]
    units = [(n, r, c) for n in range(len(wins)) for r, row in enumerate(grid) for c in range(len(row))]
    cost = [(wins[n][1] - wins[n][0]) * grid[r][c][2] * grid[r][c][3] for n, r, c in units]
    runs = balanced_runs(cost, world)
    owner = [next(rk for rk, (a, b) in enumerate(runs) if a <= u < b) for u in range(len(units))]
    per_win = len(units) // len(wins)
    wowner = [owner[n * per_win] for n in range(len(wins))]  # the rank that holds the window's first tile assembles it
    return wins, grid, units, owner, wowner


@torch.no_grad()  # (inference entry points: a module left in train() mode must not take the taped training path here)
def _units_sharded(model, x: torch.Tensor, encode: bool, gather: bool, group=None):
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    wins, grid, units, owner, wowner = unit_plan(model, x.shape, encode, world)
    net = model.encoder if encode else model.decoder
    tp = model._tile_params(encode)
    tnc, snc = model.config.time_n_compress, model.config.spatial_n_compress
    cout = [net.conv_out.weight.shape[0] if hasattr(net, "conv_out") else 0]

    def out_shape(u):  # what the network returns for unit u (receivers allocate from geometry alone)
        n, r, c = units[u]
        f = wins[n][1] - wins[n][0]
        _, _, h, w = grid[r][c]
        if encode:
            return (x.shape[0], cout[0], 1 + (f - 1) // tnc, h // snc, w // snc)
        return (x.shape[0], cout[0], 1 + (f - 1) * tnc, h * snc, w * snc)

    outs = {}
    for u, (n, r, c) in enumerate(units):
        if owner[u] != rank:
            continue
        i, j, h, w = grid[r][c]
        o = net(x[:, :, wins[n][0]:wins[n][1], i:i + h, j:j + w])
        cout[0] = cout[0] or o.shape[1]
        assert tuple(o.shape) == out_shape(u), (tuple(o.shape), out_shape(u))
        outs[u] = o
    if not hasattr(net, "conv_out"):  # (a network without the shipped layout: the ranks agree on its channel count)
        c = torch.tensor([cout[0]], device=x.device)
        dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
        cout[0] = int(c.item())
    # raw tiles to the owner of their window (both sides walk the units in the same order: FIFO per pair, tags for gloo)
    ops_p2p = []
    for u, (n, r, c) in enumerate(units):
        src, dst = owner[u], wowner[n]
        if src == dst:
            continue
        if rank == src:
            ops_p2p.append(dist.P2POp(dist.isend, outs[u].contiguous(), _peer(group, dst), group, tag=u))
            TRAFFIC["sent"] += _nbytes(outs[u])
        elif rank == dst:
            outs[u] = torch.empty(out_shape(u), dtype=x.dtype, device=x.device)
            ops_p2p.append(dist.P2POp(dist.irecv, outs[u], _peer(group, src), group, tag=u))
            TRAFFIC["recv"] += _nbytes(outs[u])
    if ops_p2p:
        for q in dist.batch_isend_irecv(ops_p2p):
            q.wait()
    # assemble my windows (blend in place in the reference's order, crop, concatenate), drop frame 0 of every window but the first
    mine = []
    per_win = len(units) // len(wins)
    for n in range(len(wins)):
        if wowner[n] != rank:
            continue
        rows, u = [], n * per_win
        for row in grid:
            rows.append([outs[u + k] for k in range(len(row))])
            u += len(row)
        o = model._assemble_tiles(rows, tp[2], tp[3]) if tp is not None else rows[0][0]
        mine.append(o if n == 0 else o[:, :, 1:])
    out = torch.cat(mine, dim=2) if mine else None
    if not gather:
        return out
    counts = []
    for rk in range(world):
        cnt = 0
        for n in range(len(wins)):
            if wowner[n] == rk:
                f = wins[n][1] - wins[n][0]
                cnt += (1 + (f - 1) // tnc if encode else 1 + (f - 1) * tnc) - (0 if n == 0 else 1)
        counts.append(cnt)
    if out is None:
        ref = out_shape(0)
        full_h = x.shape[3] // snc if encode else x.shape[3] * snc
        full_w = x.shape[4] // snc if encode else x.shape[4] * snc
        out = x.new_zeros((ref[0], ref[1], 0, full_h, full_w))
    return _gather_time(out, counts, group)


def encode_units_sharded(model, x: torch.Tensor, gather: bool = True, group=None):
    """moments of the whole clip with the (window x spatial tile) network calls split over the ranks; x: the full clip on
    every rank.  Equal to model.encode(x).latent_dist.parameters bit for bit."""
    return _units_sharded(model, x, True, gather, group)


def decode_units_sharded(model, z: torch.Tensor, gather: bool = True, group=None):
    return _units_sharded(model, z, False, gather, group)
