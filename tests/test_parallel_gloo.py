"""world_size-2 gloo test of the clip-parallel path (runs on CPU): sharding + the single
all-gather must reproduce the rank-order == clip-order of the whole batch, for equal and ragged
shards.  The forward itself is a stand-in (the HIP engine needs a GPU); the collective logic
is what is under test."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_forward(clips):
    # deterministic per-clip "logits": depends only on the clip's content
    flat = clips.reshape(clips.shape[0], -1)
    return torch.stack([flat.mean(1), flat.abs().max(1).values, flat[:, 0], flat.sum(1)], 1)


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pretorched_x_amd.parallel import clip_parallel_forward, shard_clips
    g = torch.Generator().manual_seed(123)
    clips = torch.randn(total, 3, 2, 4, 4, generator=g)          # identical on every rank
    local = shard_clips(clips)
    out = clip_parallel_forward(_fake_forward, local, total=total)
    out2 = clip_parallel_forward(_fake_forward, local, total=None)  # size-discovery path
    want = _fake_forward(clips)
    ok = torch.equal(out, want) and torch.equal(out2, want)
    q.put((rank, bool(ok), tuple(out.shape)))
    dist.barrier()
    dist.destroy_process_group()


def _run(total, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get() for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (total, 4) for _, _, shape in res), res


def test_clip_parallel_equal_shards():
    _run(total=8)


def test_clip_parallel_ragged_shards():
    _run(total=5)
