"""World-size-2 gloo run (CPU) of the multi-GPU plumbing: shard ranges, the ONE broadcast of the conditioning, and
per-sample noise that does not depend on the world size."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from reptext_amd import dist as rd

    N = 64
    spec = [("prompt_embeds", (1, 16, 32)), ("pooled", (1, 8)), ("hint0", (1, N, 128)), ("hint1", (1, N, 128)), ("mask0", (N,)), ("mask1", (N,))]
    cond = None
    if rank == 0:
        g = torch.Generator().manual_seed(0)
        r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
        cond = rd.Conditioning(r(1, 16, 32), r(1, 8), [r(1, N, 128), r(1, N, 128)], [torch.rand(N, generator=g), torch.rand(N, generator=g)])
    got = rd.broadcast_conditioning(cond, spec, "cpu")
    lo, hi = rd.shard_range(5, rank, world)
    noise = rd.sample_noise(range(lo, hi), (4, 4), 42, torch.float32, "cpu")
    q.put((rank, got.prompt_embeds.float().sum().item(), got.hints[1].float().sum().item(), got.masks[1].sum().item(), got.masks[1].dtype,
           (lo, hi), noise))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    sys.path.insert(0, ROOT)
    from reptext_amd import dist as rd

    assert [rd.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [rd.shard_range(32, r, 8) for r in range(8)] == [(4 * r, 4 * r + 4) for r in range(8)]        # C3: 4 images / GPU
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = res
    assert r0[1:5] == r1[1:5] and r0[4] == torch.float32            # identical conditioning on both ranks, masks stay fp32
    assert r0[5] == (0, 3) and r1[5] == (3, 5)
    full = rd.sample_noise(range(5), (4, 4), 42, torch.float32, "cpu")
    assert torch.equal(torch.cat([r0[6], r1[6]]), full)              # per-sample noise independent of world size
