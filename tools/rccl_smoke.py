"""First-contact smoke of the RCCL path on however many GPUs the box has (one is enough to prove that the backend loads, that
`bench.init_rccl` comes back and that the time-sharded step runs through `torch.distributed` on device tensors):
  python tools/rccl_smoke.py            (self-launches one rank per visible GPU, at most 8)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch
    import bench
    import cvvae_amd
    from cvvae_amd import dist as D
    world, rank, lr = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(lr)
    dist = bench.init_rccl(torch, world, rank, lr, seconds=120)
    torch.manual_seed(0)
    m = cvvae_amd.CVVAESD3Model()  # (the classes initialise their parameters with PyTorch's default-init statistics)
    m = m.to(torch.bfloat16).cuda().eval()
    T = bench.temporal_shard_T(world)
    xf, xl = bench.temporal_shard_input(T, 96, 128, world, rank, torch.bfloat16, "cuda")
    D.TRAFFIC.update(sent=0, recv=0)
    mom, yl = bench.temporal_shard_step(m, xl, T)
    ok = bench.temporal_shard_check(m, xf, mom, yl, world, rank, "cuda")
    print(f"rccl_smoke rank {rank}/{world}: backend {dist.get_backend()} rccl {bench._rccl_version(torch)} moments {tuple(mom.shape)} "
          f"frames {tuple(yl.shape)} sent {D.TRAFFIC['sent']} recv {D.TRAFFIC['recv']} equal_to_single_process {ok}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    assert ok


if __name__ == "__main__":
    if "WORLD_SIZE" in os.environ:
        worker()
    else:
        import subprocess
        import torch
        n = max(1, min(8, torch.cuda.device_count()))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        raise SystemExit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
                                          "--master-addr", "127.0.0.1", "--master-port", "29531", os.path.abspath(__file__)], env=env))
