"""N>1 path on CPU: world_size-2 (and 3) gloo processes shard the temporal windows, exchange the boundary frame
point-to-point and all_gather the result; must equal the single-process wrapper bit for bit.  The nets are the
test doubles of test_host_logic.py (the HIP engine needs a GPU; the sharding logic does not)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, time_sharded, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.test_host_logic import make
        from cvvae_amd import dist as D
        m = make("sd3", tile_spatial_size=144)
        g = torch.Generator().manual_seed(0)
        x = torch.rand((1, 3, T, 160, 200), generator=g) * 2 - 1
        full = m.encode(x).latent_dist.parameters
        if time_sharded:
            a, b = D.owned_frames(T, 16, world, rank)
            got = D.encode_windows_sharded(m, x[:, :, a:b].contiguous(), T_total=T, time_sharded=True)
        else:
            got = D.encode_windows_sharded(m, x)
        ok_e = torch.equal(got, full)
        z = full[:, :4]
        ref_y = m.decode(z).sample
        Tz = z.shape[2]
        if time_sharded:
            a, b = D.owned_frames(Tz, 4, world, rank)
            y = D.decode_windows_sharded(m, z[:, :, a:b].contiguous(), T_total=Tz, time_sharded=True)
        else:
            y = D.decode_windows_sharded(m, z)
        ok_d = torch.equal(y, ref_y)
        q.put((rank, ok_e, ok_d, tuple(got.shape), tuple(y.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T,time_sharded", [(2, 33, False), (2, 33, True), (2, 49, True), (3, 33, True), (2, 17, True)])
def test_window_sharding_matches_single_process(world, T, time_sharded):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, time_sharded, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_e, ok_d, se, sy in res:
        assert ok_e and ok_d, (rank, se, sy)


def test_partition_helpers():
    from cvvae_amd import dist as D
    assert [D.split_contiguous(8, 8, r) for r in range(8)] == [(r, r + 1) for r in range(8)]
    assert [D.split_contiguous(3, 2, r) for r in range(2)] == [(0, 2), (2, 3)]
    # cfg 4: T=129 over 8 ranks -> rank 0 owns frames 0..16, rank r owns 16r+1..16r+16
    assert [D.owned_frames(129, 16, 8, r) for r in range(8)] == [(0, 17)] + [(16 * r + 1, 16 * r + 17) for r in range(1, 8)]
    assert D.owned_frames(17, 16, 2, 1) == (0, 0)  # fewer windows than ranks


def _worker_real(rank, world, port, q):
    """the REAL CVVAESD3Model (full-size networks) sharded over gloo ranks, its kernels emulated on the CPU (tests/emu_ops.py)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cvvae_amd
        from cvvae_amd import dist as D
        from oracle.seeded import seeded_input, seeded_state_dict
        from tests import emu_ops
        torch.set_num_threads(4)
        m = cvvae_amd.CVVAESD3Model()
        m.load_state_dict(seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 0), strict=True)
        m = m.eval()
        x = seeded_input((1, 3, 33, 32, 32), 2)
        with emu_ops.patched(whole_model=True), torch.no_grad():
            full = m.encode(x).latent_dist.parameters
            a, b = D.owned_frames(33, 16, world, rank)
            got = D.encode_windows_sharded(m, x[:, :, a:b].contiguous(), T_total=33, time_sharded=True)
            z = full[:, :16]
            ref_y = m.decode(z).sample
            a, b = D.owned_frames(z.shape[2], 4, world, rank)
            y = D.decode_windows_sharded(m, z[:, :, a:b].contiguous(), T_total=z.shape[2], time_sharded=True)
        q.put((rank, torch.equal(got, full), torch.equal(y, ref_y), tuple(got.shape), tuple(y.shape)))
    finally:
        dist.destroy_process_group()


def test_real_model_window_sharding_on_emulated_kernels():
    """cfg 4's partition on the real model classes: two ranks, one 17-frame window each, the boundary frame exchanged point-to-point,
    results all-gathered -- equal to the single-process wrapper bit for bit"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_e, ok_d, se, sy in res:
        assert ok_e and ok_d, (rank, se, sy)


def _worker_units(rank, world, port, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.test_host_logic import make
        from cvvae_amd import dist as D
        m = make("sd3", tile_spatial_size=144)
        g = torch.Generator().manual_seed(0)
        x = torch.rand((1, 3, T, 160, 200), generator=g) * 2 - 1
        full = m.encode(x).latent_dist.parameters
        wins, grid, units, owner, wowner = D.unit_plan(m, x.shape, True, world)
        got = D.encode_units_sharded(m, x)
        z = full[:, :4]
        ref_y = m.decode(z).sample
        y = D.decode_units_sharded(m, z)
        mine = D.encode_units_sharded(m, x, gather=False)
        q.put((rank, torch.equal(got, full), torch.equal(y, ref_y), len(units), sorted(set(owner)), None if mine is None else mine.shape[2]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T", [(4, 33), (4, 17), (3, 49), (2, 5)])
def test_window_x_tile_units_match_single_process(world, T):
    """SURVEY 8(e) "windows x tiles": a 160x200 clip with 144-pixel tiles is a 2x2 tile grid per window; the (window, tile) network
    calls are split over the ranks in contiguous area-balanced runs, raw tiles go point-to-point to the rank that assembles the
    window, results are all-gathered: equal to the single-process wrapper bit for bit -- including a ONE-window clip (T = 17, 5)
    whose four tiles keep four ranks busy, which the window partition cannot do."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_units, args=(r, world, port, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_e, ok_d, n_units, owners, _ in res:
        assert ok_e and ok_d, rank
        nwin = max(1, -(-(T - 1) // 16))
        assert n_units == 4 * nwin
        assert len(owners) == min(world, n_units), owners  # every rank that can have work has some


def test_unit_plan_cfg4_and_short_clips():
    """cfg 4 (129 frames, 720x1280): 8 windows x 6 tiles = 48 units, 6 per rank on 8 ranks, every window assembled by the rank that
    computes it (no tile traffic).  A 17-frame 720p clip: 6 units on 6 of 8 ranks; a 65-frame one: all 8 ranks busy."""
    import cvvae_amd
    from cvvae_amd import dist as D
    m = cvvae_amd.CVVAESD3Model.__new__(cvvae_amd.CVVAESD3Model)
    torch.nn.Module.__init__(m)
    m._init_common({})
    wins, grid, units, owner, wowner = D.unit_plan(m, (1, 3, 129, 720, 1280), True, 8)
    assert len(wins) == 8 and [len(r) for r in grid] == [3, 3] and len(units) == 48
    assert [owner.count(r) for r in range(8)] == [6] * 8 and wowner == list(range(8))
    assert all(owner[u] == wowner[units[u][0]] for u in range(48))
    _, _, units, owner, wowner = D.unit_plan(m, (1, 3, 17, 720, 1280), True, 8)
    assert len(units) == 6 and len(set(owner)) == 6 and len(set(wowner)) == 1
    _, _, units, owner, _ = D.unit_plan(m, (1, 3, 65, 720, 1280), True, 8)
    assert len(units) == 24 and len(set(owner)) == 8
    # decode side: latent tiles of 72 with stride 56
    wins, grid, units, owner, _ = D.unit_plan(m, (1, 16, 33, 90, 160), False, 8)
    assert len(wins) == 8 and [len(r) for r in grid] == [3, 3] and [owner.count(r) for r in range(8)] == [6] * 8


def _worker_bench_step(rank, world, port, q):
    """bench.py's own N > 1 functions -- temporal_shard_input / temporal_shard_step / temporal_shard_check -- on the REAL
    CVVAESD3Model over gloo (kernels emulated on the CPU): one clip of 1 + 16 N frames, time-sharded input, halo by batched
    send/recv, moments all-gathered, every rank checking its own slice against the single-process wrapper"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        import cvvae_amd
        from cvvae_amd import dist as D
        from oracle.seeded import seeded_state_dict
        from tests import emu_ops
        torch.set_num_threads(4)
        m = cvvae_amd.CVVAESD3Model()
        m.load_state_dict(seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 0), strict=True)
        m = m.eval()
        T = bench.temporal_shard_T(world)
        x_full, x_local = bench.temporal_shard_input(T, 32, 32, world, rank, torch.float32, "cpu")
        a, b = D.owned_frames(T, 16, world, rank)
        assert x_local.shape[2] == b - a == (17 if rank == 0 else 16)
        with emu_ops.patched(whole_model=True), torch.no_grad():
            D.TRAFFIC.update(sent=0, recv=0)
            mom, y_local = bench.temporal_shard_step(m, x_local, T)
            sent, recv = D.TRAFFIC["sent"], D.TRAFFIC["recv"]
            ok = bench.temporal_shard_check(m, x_full, mom, y_local, world, rank, "cpu")
        frame = x_local[:, :, :1].numel() * 4
        gathered = 2 * (mom.numel() // mom.shape[2]) * 5 * 4 * (world - 1) // 2  # padded all_gather: 5 latent frames per rank
        q.put((rank, ok, tuple(mom.shape), tuple(y_local.shape), sent, recv, frame, gathered))
    finally:
        dist.destroy_process_group()


def test_bench_temporal_shard_step_over_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_worker_bench_step, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ms, ys, sent, recv, frame, gathered in res:
        assert ok, (rank, ms, ys)
        assert ms[2] == 9 and ys[2] == (17 if rank == 0 else 16)
        # wire accounting: the halo frame goes 0 -> 1 (one pixel frame), the gather moves every other rank's padded moments
        assert sent == (frame if rank == 0 else 0) + gathered and recv == (frame if rank == 1 else 0) + gathered, (rank, sent, recv)


def _worker_subgroup(rank, world, port, q):
    """the sharding entry points on a SUBGROUP (ranks 1..3 of 4): P2POp peers are global ranks whatever the group is"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.test_host_logic import make
        from cvvae_amd import dist as D
        grp = dist.new_group([1, 2, 3])
        if rank == 0:
            q.put((rank, True, True))
            return
        gr, gw = dist.get_rank(grp), dist.get_world_size(grp)
        m = make("sd3", tile_spatial_size=144)
        g = torch.Generator().manual_seed(0)
        x = torch.rand((1, 3, 49, 160, 200), generator=g) * 2 - 1
        full = m.encode(x).latent_dist.parameters
        a, b = D.owned_frames(49, 16, gw, gr)
        got = D.encode_windows_sharded(m, x[:, :, a:b].contiguous(), T_total=49, time_sharded=True, group=grp)
        x1 = x[:, :, :17].contiguous()  # one window, 2x2 tiles over three ranks: raw tiles point-to-point inside the subgroup
        got_u = D.encode_units_sharded(m, x1, group=grp)
        q.put((rank, torch.equal(got, full), torch.equal(got_u, m.encode(x1).latent_dist.parameters)))
    finally:
        dist.destroy_process_group()


def test_sharding_on_a_subgroup():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_subgroup, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(a and b for _, a, b in res), res


def _worker_ddp(rank, world, port, q):
    """the trainable encoder (kernels emulated on the CPU) under DistributedDataParallel, as the reference's trainer wraps its
    autoencoder (`strategy: ddp`): each rank a different clip; the gradients DDP leaves in .grad must be the MEAN over the ranks
    of the single-process gradients -- i.e. the parameter gradients our autograd nodes return go through AccumulateGrad and DDP's
    bucket all-reduce like any other"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cvvae_amd
        from oracle.seeded import seeded_input, seeded_state_dict
        from tests import emu_ops
        torch.set_num_threads(2)
        small = dict(block_out_channels=[128, 256], layers_per_block=1)
        m = cvvae_amd.CVVAESD3Model(**small)
        m.load_state_dict(seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 7), strict=True)
        enc = m.encoder.train()
        xs = [seeded_input((1, 3, 3, 8, 8), 20 + r) for r in range(world)]
        with emu_ops.patched(whole_model=True):
            # single-process reference: gradient of each rank's loss, averaged
            ref = {n: torch.zeros_like(p) for n, p in enc.named_parameters()}
            for xr in xs:
                enc.zero_grad(set_to_none=True)
                enc(xr).float().pow(2).mean().backward()
                for n, p in enc.named_parameters():
                    ref[n] += p.grad / world
            enc.zero_grad(set_to_none=True)
            ddp = torch.nn.parallel.DistributedDataParallel(enc)
            ddp(xs[rank]).float().pow(2).mean().backward()
            worst = max(float((p.grad - ref[n]).abs().max() / ref[n].abs().max().clamp_min(1e-20)) for n, p in enc.named_parameters())
            have = all(p.grad is not None for p in enc.parameters())
        q.put((rank, have, worst))
    finally:
        dist.destroy_process_group()


def test_trainable_encoder_under_ddp_over_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ddp, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, have, worst in res:
        assert have and worst < 1e-5, (rank, have, worst)
