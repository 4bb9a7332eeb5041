"""Read-sharded multi-GPU merge.

Reads are independent, so each rank (one process per GPU, store replicated in its HBM) classifies its own
slice of the read stream with no data-path collective.  One exchange at the end of a run merges the
accumulators over RCCL/xGMI (backend "nccl" on ROCm) -- or gloo on CPU tensors in the tests:

* additive per-taxid columns          -> all_reduce(SUM)   int64 [n_values x GS_N_SUMS]
* (maxContigLen << 40 | ~readNo) keys -> all_reduce(MAX)   int64 [n_values]  (first read with the max wins,
                                                           which is what a single-threaded run records)
* double error sums                   -> all_reduce(SUM)   (order dependent, not part of the bit-exact contract)
* unique-k-mer bitmap                 -> all_gather + OR   (RCCL has no bitwise-OR reduction)

All tensors are plain torch tensors (views of the library's device buffers in bench.py).
"""
import torch
import torch.distributed as dist


def merge_run_state(sums, max_keys, dsums, bitmap, group=None, or_parts=None, force=False):
    """In-place merge over the process group; afterwards every rank holds the global state.

    or_parts(gathered, world): optional hook that ORs `world` back-to-back bitmaps into the run's bitmap
    (bench.py passes gs_match_or_bitmap); the default does it with torch ops.
    force: run the collectives even for a single-rank group (rehearsal of the multi-GPU path on one GPU).
    """
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_keys, op=dist.ReduceOp.MAX, group=group)
    if dsums is not None:
        dist.all_reduce(dsums, op=dist.ReduceOp.SUM, group=group)
    gathered = torch.empty(world * bitmap.numel(), dtype=bitmap.dtype, device=bitmap.device)
    dist.all_gather_into_tensor(gathered, bitmap.contiguous(), group=group)
    if or_parts is not None:
        if bitmap.is_cuda:
            torch.cuda.synchronize(bitmap.device)
        or_parts(gathered, world)
    else:
        parts = gathered.view(world, -1)
        acc = parts[0].clone()
        for i in range(1, world):
            acc |= parts[i]
        bitmap.copy_(acc)


def shard_bounds(n_total, rank, world):
    """contiguous read range [lo, hi) of `rank` (global readNo is kept)"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
