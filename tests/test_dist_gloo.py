"""World-size-2 gloo run (CPU) of the multi-GPU plumbing bench.py uses for N > 1: shard ranges, the ONE broadcast of the
conditioning (one prompt / hint / mask per image, or one shared by the batch), each rank's rows, and per-sample noise.
Host-logic level: what every rank would feed its replica equals, row for row and bit for bit, what a single process feeds
for the same global batch — so a sample's result cannot depend on the world size."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, L, D, N = 5, 16, 32, 64           # global batch, text tokens, joint dim, image tokens (small stand-ins)


def _conditioning(shared: bool):
    from reptext_amd import dist as rd

    g = torch.Generator().manual_seed(0)
    Gc = 1 if shared else G
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    return rd.Conditioning(r(Gc, L, D), r(Gc, 8), [r(Gc, N, 128), r(Gc, N, 128)], [torch.rand(Gc, N, generator=g), torch.rand(Gc, N, generator=g)])


def _worker(rank, world, port, q, shared):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from reptext_amd import dist as rd

    Gc = 1 if shared else G
    spec = [("prompt_embeds", (Gc, L, D)), ("pooled", (Gc, 8)), ("hint0", (Gc, N, 128)), ("hint1", (Gc, N, 128)), ("mask0", (Gc, N)), ("mask1", (Gc, N))]
    cond = _conditioning(shared) if rank == 0 else None
    got = rd.broadcast_conditioning(cond, spec, "cpu")                   # the ONE collective of the path
    lo, hi = rd.shard_range(G, rank, world)
    mine = got.shard(lo, hi)
    noise = rd.sample_noise(range(lo, hi), (4, 4), 42, torch.float32, "cpu")
    # numpy copies: pickled by value (torch tensors would travel as shared-memory handles that die with this process)
    npy = lambda t: t.contiguous().float().numpy().copy()
    q.put((rank, (lo, hi), npy(mine.prompt_embeds), npy(mine.pooled), [npy(h) for h in mine.hints], [npy(m) for m in mine.masks], npy(noise),
           str(got.masks[1].dtype), str(got.prompt_embeds.dtype)))
    dist.barrier()
    dist.destroy_process_group()


def _run_world2(shared):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if shared else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, shared)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_broadcast_and_sharding_world2():
    sys.path.insert(0, ROOT)
    from reptext_amd import dist as rd

    assert [rd.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [rd.shard_range(32, r, 8) for r in range(8)] == [(4 * r, 4 * r + 4) for r in range(8)]        # C3: 4 images / GPU
    assert [rd.shard_range(32, r, 1) for r in range(1)] == [(0, 32)]                                      # strong scaling, N = 1
    for shared in (False, True):
        r0, r1 = _run_world2(shared)
        assert r0[1] == (0, 3) and r1[1] == (3, 5)
        assert r0[7] == "torch.float32" and r0[8] == "torch.bfloat16"                 # masks stay fp32, embeddings bf16
        # what a single process (world 1) feeds for the same global batch
        full = _conditioning(shared).shard(0, G)
        cat = lambda i: torch.cat([torch.from_numpy(r0[i]), torch.from_numpy(r1[i])])
        assert torch.equal(cat(2).float(), full.prompt_embeds.contiguous()) and torch.equal(cat(3).float(), full.pooled.contiguous())
        for line in range(2):
            assert torch.equal(torch.cat([torch.from_numpy(r0[4][line]), torch.from_numpy(r1[4][line])]), full.hints[line].contiguous())
            assert torch.equal(torch.cat([torch.from_numpy(r0[5][line]), torch.from_numpy(r1[5][line])]), full.masks[line].contiguous())
        assert torch.equal(cat(6), rd.sample_noise(range(G), (4, 4), 42, torch.float32, "cpu"))   # noise by global sample id


def test_conditioning_shard_rules():
    sys.path.insert(0, ROOT)
    import pytest

    from reptext_amd import dist as rd

    c = _conditioning(False)
    s = c.shard(1, 4)
    assert s.prompt_embeds.shape[0] == 3 and torch.equal(s.hints[1], c.hints[1][1:4]) and torch.equal(s.masks[0], c.masks[0][1:4])
    sh = _conditioning(True).shard(2, 5)                                   # shared: expanded, not copied
    assert sh.prompt_embeds.shape[0] == 3 and sh.prompt_embeds.stride(0) == 0 and sh.masks[1].shape == (3, N)
    with pytest.raises(ValueError):
        c.shard(3, 9)
    legacy = rd.Conditioning(c.prompt_embeds[:1], c.pooled[:1], [c.hints[0][:1]], [torch.rand(N)])   # [N] masks of round 1
    assert legacy.masks[0].shape == (1, N)
