"""Clip-parallel inference: one process per GPU, clips sharded on the batch dimension, weights
replicated, ONE all-gather of the fp32 logits per forward (RCCL over xGMI when the process group
uses the "nccl" backend; gloo on CPU for the ordering tests).

This is the MI355X-native counterpart of the reference's only multi-GPU mechanism,
`torch.nn.DataParallel(model)` (reference examples/imagenet_eval.py:136, nonlocalnet.py:604):
same dim-0 contiguous chunking and the same rank-order == clip-order gather, but with no
per-step weight broadcast and no single-process GIL bottleneck.  The payload is tiny
(8 x 339 fp32 = 10.8 KB per rank for config 2), i.e. latency bound: a single collective is the
right shape for it -- no bucketing, no ring tuning.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous chunk [start, end) of `total` clips owned by `rank` -- torch.chunk semantics
    (ceil-sized leading chunks), i.e. what DataParallel's scatter does."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad rank/world_size")
    per = -(-total // world_size)
    start = min(rank * per, total)
    return start, min(start + per, total)


def shard_clips(clips, world_size=None, rank=None):
    """This rank's clips out of a globally identical batch tensor."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    a, b = shard_bounds(clips.shape[0], world_size, rank)
    return clips[a:b]


def gather_logits(local_logits, total=None, group=None):
    """All-gather per-rank logits [B_r, classes] into [sum B_r, classes], rank order == clip order.

    Equal shards use one `all_gather_into_tensor`; ragged shards (total not divisible by the world
    size) are padded to the largest shard and trimmed afterwards.
    """
    world = dist.get_world_size(group)
    if world == 1:
        return local_logits
    classes = local_logits.shape[1]
    if total is None:
        sizes = [torch.zeros(1, dtype=torch.int64, device=local_logits.device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([local_logits.shape[0]], dtype=torch.int64,
                                            device=local_logits.device), group=group)
        counts = [int(s.item()) for s in sizes]
    else:
        counts = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
    mx = max(counts)
    if local_logits.shape[0] != mx:
        pad = local_logits.new_zeros((mx - local_logits.shape[0], classes))
        local_logits = torch.cat([local_logits, pad], 0)
    out = local_logits.new_empty((world * mx, classes))
    dist.all_gather_into_tensor(out, local_logits.contiguous(), group=group)
    if all(c == mx for c in counts):
        return out
    return torch.cat([out[r * mx: r * mx + c] for r, c in enumerate(counts)], 0)


def clip_parallel_forward(forward_fn, local_clips, total=None, group=None):
    """Run `forward_fn` (e.g. a pretorched_x_amd model) on this rank's clips and return the logits
    of the WHOLE batch on every rank."""
    return gather_logits(forward_fn(local_clips), total=total, group=group)


def verify_gather(local_logits, gathered, group=None):
    """Self-check of the clip-parallel step, run on EVERY rank (bench.py does it after the timed region):
      * order: this rank's rows of the gathered tensor are bit-identical to what it computed
        (rank order == clip order, the reference's DataParallel gather semantics,
        examples/imagenet_eval.py:136);
      * replicas: every rank holds bit-identical gathered logits (element-wise MAX == MIN over ranks of the
        int32 bit patterns);
    both flags are all-reduced, so the returned dict is the same on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = local_logits.shape[0]
    counts = torch.zeros(world, dtype=torch.int64, device=local_logits.device)
    counts[rank] = n
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    start = int(counts[:rank].sum().item())
    total = int(counts.sum().item())
    order_ok = gathered.shape[0] == total and bool(torch.equal(gathered[start:start + n], local_logits))
    bits = gathered.contiguous().view(torch.int32)
    hi, lo = bits.clone(), bits.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    same = bool(torch.equal(hi, lo))
    flags = torch.tensor([int(order_ok), int(same)], dtype=torch.int32, device=local_logits.device)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN, group=group)
    return {"gather_order_ok": bool(flags[0].item()), "replicas_identical": bool(flags[1].item()),
            "ranks": world, "rows": total}


def broadcast_tuned_table(src=0, group=None):
    """Tile choices measured on rank `src` (Engine.autotune) adopted by every rank: N concurrent tuners on one
    node perturb each other's HIP-event timings, and every rank should run the same kernels anyway."""
    from . import engine as _engine
    box = [_engine.tuned_snapshot() if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    if dist.get_rank(group) != src:
        _engine.tuned_merge(box[0])
    return len(box[0])
