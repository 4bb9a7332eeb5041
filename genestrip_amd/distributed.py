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


class _DevArray:
    """device memory of the library as a torch tensor (no copy)"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": (n,), "typestr": typestr, "version": 2}


def match_files_sharded(matcher, paths, group=None, device=None, via_host=False):
    """runMatcher over the files of a sample with one process per GPU: rank r takes files r, r + world, ...
    (gs_host_match_into: raw text blocks to its own GPU), then the runs are merged (merge_run_state) and finished.
    Returns (table, dtable, (reads, kmers, bps)) -- identical on every rank and identical to a single-process run over
    the same files in the same order.  via_host: merge host copies (gloo groups); default: the device buffers over RCCL."""
    import numpy as np
    from . import binding as _b
    from . import host as _h

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    paths = list(paths)
    if len(paths) > 256:  # gs_host_match_into: (file << 32 | read) must fit the 40-bit read-number field
        raise ValueError("match_files_sharded handles at most 256 files per run")
    mine = list(range(rank, len(paths), world))
    counts_mine, tot = _h.match_files_into(matcher, [paths[i] for i in mine], mine)
    st = matcher.device_state()  # syncs the run's stream and refreshes the compact unique bitmap
    nv = matcher.store.n_values
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    t_sums = torch.as_tensor(_DevArray(st["sums"], nv * _b.N_SUMS, "<i8"), device=dev)
    t_max = torch.as_tensor(_DevArray(st["max_keys"], nv, "<i8"), device=dev)
    t_dsum = torch.as_tensor(_DevArray(st["dsums"], nv * _b.N_DCOLS, "<f8"), device=dev)
    t_bits = torch.as_tensor(_DevArray(st["bitmap"], st["bitmap_words"], "<i4"), device=dev)
    counts = torch.zeros(len(paths) + 3, dtype=torch.int64)
    for i, c in zip(mine, counts_mine):
        counts[i] = int(c)
    counts[len(paths):] = torch.tensor([tot.reads, tot.kmers, tot.bps])
    if via_host:
        hs, hm, hd, hb = t_sums.cpu(), t_max.cpu(), t_dsum.cpu(), t_bits.cpu()
        merge_run_state(hs, hm, hd, hb, group=group)
        t_sums.copy_(hs), t_max.copy_(hm), t_dsum.copy_(hd), t_bits.copy_(hb)
        torch.cuda.synchronize(dev)
        matcher.or_bitmap(t_bits.data_ptr(), 1)
        dist.all_reduce(counts, group=group)
    else:
        merge_run_state(t_sums, t_max, t_dsum, t_bits, group=group, or_parts=lambda g, w: matcher.or_bitmap(g.data_ptr(), w))
        cd = counts.to(dev)
        dist.all_reduce(cd, group=group)
        counts = cd.cpu()
    table, dtable = matcher.finish()
    # (file << 32 | read in file) -> running read number over the files in order
    before = np.concatenate([[0], np.cumsum(counts[:len(paths)].numpy())])
    col = table[:, _b.N_COLS - 1]
    has = col >= 0
    col[has] = before[col[has] >> 32] + (col[has] & 0xFFFFFFFF)
    return table, dtable, tuple(int(x) for x in counts[len(paths):])


def _attach_all(store, rank, world, group):
    handles = [None] * world
    dist.all_gather_object(handles, store.export_stripe(), group=group)
    for q in range(world):
        if q != rank:
            store.attach_stripe(q, handles[q])
    dist.barrier(group)  # nobody runs before every stripe is attached everywhere, nobody frees before that either
    return store


def striped_store_close(store, matchers=(), group=None):
    """Collective.  Takes a striped store down in the only safe order: every rank finishes (closes) its runs, ALL ranks meet,
    then every rank frees its handle -- which unmaps the foreign stripes and frees its own.  Without the barrier a rank could
    free its stripe while a kernel of another rank still reads record lines from it through the IPC mapping (a GPU memory fault
    in that process)."""
    for m in matchers:
        m.sync()
        m.close()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group)
    store.close()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group)


def striped_store_from_file(path, device=0, group=None):
    """Collective.  As striped_store, from a store file (DeviceKMerStore.save of the store built once): every rank reads the
    image, keeps its stripe and attaches the others; nobody rebuilds the layout."""
    from .binding import DeviceKMerStore
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return DeviceKMerStore.load(path, device=device)
    if world > 8:
        raise ValueError("a striped store spans the GPUs of one node (at most 8 ranks)")
    rank = dist.get_rank(group)
    return _attach_all(DeviceKMerStore.load_stripe(path, device=device, n_stripes=world, stripe=rank), rank, world, group)


def striped_store(k, kmers, value_idx, n_values, parent_vi=None, device=0, group=None):
    """Collective.  ONE store over the HBM of all ranks' GPUs (include/gsgpu.h, "striped store"): every rank builds the
    layout from the same arrays and keeps stripe `rank` of the super-k-mer record table (gs_db_create_stripe), exports it
    as a HIP IPC memory handle, and attaches the stripes of the others.  The returned store serves the ordinary fused path
    (FastqKMerMatcher.submit*): record lines of foreign stripes are loaded over xGMI, there is no exchange step per batch.
    The runs of the ranks merge like runs on replicas (merge_run_state).  One rank: a plain store."""
    from .binding import DeviceKMerStore
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return DeviceKMerStore(k, kmers, value_idx, n_values, parent_vi, device=device)
    if world > 8:
        raise ValueError("a striped store spans the GPUs of one node (at most 8 ranks)")
    rank = dist.get_rank(group)
    store = DeviceKMerStore.stripe(k, kmers, value_idx, n_values, parent_vi, device=device, n_stripes=world, stripe=rank)
    return _attach_all(store, rank, world, group)


def shard_bounds(n_total, rank, world):
    """contiguous read range [lo, hi) of `rank` (global readNo is kept)"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------------------------
# DB-partitioned mode (BASELINE.json configs[4]): the store is range-split over the ranks by key
# hash, every k-mer of a read is routed to the rank that owns it (all-to-all over xGMI), probed there, and the node
# comes back to the read's home rank, which runs the per-read reduce.  Unique k-mers are marked in the owner's table,
# so the per-rank unique counts are disjoint and add up.
# ---------------------------------------------------------------------------------------------------------------
OWNER_SHIFT = 40          # genestrip_amd/csrc/gs_layout.h: GS_OWNER_SHIFT
KEY_INVALID = -1          # ~0 as int64: window with a non-CGAT base, never routed
KEY_MISS = -2             # ~0 - 1: the store-wide minimizer gate rules the k-mer out, never routed (node = miss)
NODE_MISS = -1
NODE_INVALID = -2


def position_offsets(offsets, k):
    """exclusive prefix of max(0, L - k + 1) over the reads of a batch (int64 tensor, n + 1 entries)"""
    lens = offsets[1:] - offsets[:-1]
    cnt = torch.clamp(lens - (k - 1), min=0)
    out = torch.zeros(offsets.numel(), dtype=torch.int64, device=offsets.device)
    torch.cumsum(cnt, 0, out=out[1:])
    return out


def plan_routing(keys, world):
    """keys: int64 tensor of mixed keys (KEY_INVALID for invalid windows, KEY_MISS for gate-rejected k-mers; real
    keys are < 2^62, i.e. non-negative).
    Returns (idx, send_keys, counts): idx = positions of the routed keys in owner-sorted order, send_keys =
    keys[idx] (grouped by owner rank, ascending), counts[j] = number of keys going to rank j."""
    valid = keys >= 0
    idx_valid = torch.nonzero(valid, as_tuple=False).flatten()
    owner = (keys[idx_valid] >> OWNER_SHIFT) % world
    order = torch.argsort(owner, stable=True)
    idx = idx_valid[order]
    counts = torch.bincount(owner, minlength=world)
    return idx, keys[idx].contiguous(), counts


def scatter_nodes(nodes_sorted, idx, n_keys, keys=None):
    """inverse of plan_routing for the returned nodes; unrouted positions read NODE_INVALID, or NODE_MISS where
    keys (the tensor plan_routing saw) holds KEY_MISS"""
    nodes = torch.full((n_keys,), NODE_INVALID, dtype=torch.int32, device=nodes_sorted.device)
    if keys is not None:
        nodes[keys[:n_keys] == KEY_MISS] = NODE_MISS
    nodes[idx] = nodes_sorted
    return nodes


def exchange_all_to_all(send, send_counts, group=None):
    """variable-size all-to-all of a 1-D tensor: returns (recv, recv_counts)"""
    world = dist.get_world_size(group)
    sc = send_counts.to(torch.int64)
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    recv = torch.empty(int(rc.sum().item()), dtype=send.dtype, device=send.device)
    dist.all_to_all_single(recv, send, output_split_sizes=rc.tolist(), input_split_sizes=sc.tolist(), group=group)
    assert len(rc) == world
    return recv, rc


ROUTE_CHUNK = 2048  # genestrip_amd/csrc/gs_params.h: GS_ROUTE_CHUNK (the library reports it: FastqKMerMatcher.route_geometry)
ROUTE_OVERFLOWS = [0]  # batches of partitioned_match_batch that fell back to the unfused steps


def _all_to_all_views(recv, recv_counts, send_views, send_counts, group):
    """variable-size all-to-all whose send side is a list of views (one region per destination).  RCCL takes the list as
    it is (grouped send / recv, no staging copy); other backends (gloo in the rehearsal tests) get one contiguous buffer"""
    if dist.get_backend(group) == "nccl":
        dist.all_to_all(list(torch.split(recv, recv_counts)), send_views, group=group)
    else:
        send = torch.cat(send_views) if send_views else recv.new_empty(0)
        dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)


MAX_BATCH_POSITIONS = 0xfffffffe  # a routed key carries its position in the batch as 32 bits, ~0 marks an unused slot


def check_batch_positions(n_keys):
    """the k-mer positions of one DB-partitioned batch must fit the 32-bit routing index (gs_match_encode_route and gs_route_keys
    refuse more as well): ~35.8 M reads of 150 bp at k = 31 -- split larger batches"""
    if n_keys > MAX_BATCH_POSITIONS:
        raise ValueError("a DB-partitioned batch holds %d k-mer positions; at most %d fit the routing index: split the batch"
                         % (n_keys, MAX_BATCH_POSITIONS))
    return n_keys


def partitioned_match_batch(matcher, k, seq, offsets, n_reads, first_read_no=0, group=None, class_vi=None, flags=None,
                            cap=None):
    """One batch in DB-partitioned mode on this rank (matcher's store = this rank's partition).
    seq / offsets: device tensors (uint8 / int64).  Collective: every rank of the group must call it.
    cap: slots per owner region (tests; default: sized from the batch).

    encode + routing in one kernel (gs_match_encode_route: the keys go straight into one send region per owner, in
    chunks of 2048 slots, unused slots carry sentinels) -> all-to-all of the regions -> probe on the owner -> all-to-all
    back -> scatter per region -> reduce.  If a region overflows (keys that all hash to one owner: degenerate input)
    the batch goes through the unfused steps (partitioned_match_batch_unfused)."""
    world = dist.get_world_size(group)
    dev = seq.device
    pos_off = position_offsets(offsets[:n_reads + 1], k)
    n_keys = check_batch_positions(int(pos_off[-1].item()))
    if n_keys == 0:
        nodes = torch.empty(1, dtype=torch.int32, device=dev)
        matcher.reduce(seq, offsets, pos_off, nodes, n_reads, first_read_no, class_vi, flags)
        matcher.sync()
        return
    # room per owner: its fair share of ALL positions with a quarter on top (only gate-passing k-mers are routed, and a
    # chunk is given up when it cannot take the keys of a sub-round) + one chunk per wave that may stay partly used
    n_waves, chunk = matcher.route_geometry(n_reads)  # (the library's launch geometry on this device, not a copy of it)
    if cap is None:
        cap = n_keys // world + n_keys // (4 * world) + (n_waves + 2) * chunk
    cap = (cap + chunk - 1) // chunk * chunk
    send_keys = torch.empty(world * cap, dtype=torch.int64, device=dev)
    send_idx = torch.empty(world * cap, dtype=torch.int32, device=dev)
    nodes = torch.empty(n_keys, dtype=torch.int32, device=dev)
    counts, overflow = matcher.encode_route(seq, offsets, pos_off, n_reads, world, cap, send_keys, send_idx, nodes)
    flag = torch.tensor([int(overflow)], dtype=torch.int64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)  # every rank takes the same route
    if int(flag.item()):
        ROUTE_OVERFLOWS[0] += 1  # (a batch whose regions were too small pays for the fused encode AND the unfused steps: worth noticing)
        return partitioned_match_batch_unfused(matcher, k, seq, offsets, n_reads, first_read_no, group, class_vi, flags)
    sc = torch.tensor(counts, dtype=torch.int64, device=dev)
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    rcl = rc.tolist()
    send_list = [send_keys[o * cap:o * cap + counts[o]] for o in range(world)]
    n_recv = sum(rcl)
    recv_keys = torch.empty(max(n_recv, 1), dtype=torch.int64, device=dev)
    _all_to_all_views(recv_keys[:n_recv], rcl, send_list, counts, group)
    recv_nodes = torch.empty(max(n_recv, 1), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    matcher.probe_keys(recv_keys, recv_nodes, n_recv)
    matcher.sync()
    back = torch.empty(max(sum(counts), 1), dtype=torch.int32, device=dev)
    back_list = list(torch.split(back[:sum(counts)], counts))
    _all_to_all_views(back[:sum(counts)], counts, list(torch.split(recv_nodes[:n_recv], rcl)), rcl, group)
    torch.cuda.synchronize(dev)
    for o in range(world):
        matcher.unroute_region(send_idx[o * cap:o * cap + counts[o]], back_list[o], counts[o], nodes)
    matcher.reduce(seq, offsets, pos_off, nodes, n_reads, first_read_no, class_vi, flags)
    matcher.sync()


def partitioned_match_batch_unfused(matcher, k, seq, offsets, n_reads, first_read_no=0, group=None, class_vi=None, flags=None):
    """One batch in DB-partitioned mode on this rank (matcher's store = this rank's partition).
    seq / offsets: device tensors (uint8 / int64).  Collective: every rank of the group must call it."""
    world = dist.get_world_size(group)
    pos_off = position_offsets(offsets[:n_reads + 1], k)
    n_keys = check_batch_positions(int(pos_off[-1].item()))
    keys = torch.empty(max(n_keys, 1), dtype=torch.int64, device=seq.device)
    matcher.encode(seq, offsets, pos_off, keys, n_reads)
    matcher.sync()
    keys = keys[:n_keys]
    # group the valid keys by owner rank: counting sort on the device (gs_route_keys); plan_routing / scatter_nodes
    # are the same steps written with torch ops (used by the CPU/gloo tests)
    send_buf = torch.empty(max(n_keys, 1), dtype=torch.int64, device=seq.device)
    idx = torch.empty(max(n_keys, 1), dtype=torch.int32, device=seq.device)
    nodes = torch.empty(max(n_keys, 1), dtype=torch.int32, device=seq.device)
    counts = torch.tensor(matcher.route_keys(keys, n_keys, world, send_buf, idx, nodes), dtype=torch.int64, device=seq.device)
    n_routed = int(counts.sum().item())
    recv_keys, recv_counts = exchange_all_to_all(send_buf[:n_routed], counts, group)
    recv_nodes = torch.empty(max(recv_keys.numel(), 1), dtype=torch.int32, device=seq.device)
    torch.cuda.synchronize(seq.device)
    matcher.probe_keys(recv_keys, recv_nodes, recv_keys.numel())
    matcher.sync()
    back, _ = exchange_all_to_all(recv_nodes[:recv_keys.numel()].contiguous(), recv_counts, group)
    torch.cuda.synchronize(seq.device)
    matcher.unroute_nodes(None, idx, back, n_routed, nodes, n_keys)  # (route_keys wrote the unrouted positions)
    matcher.reduce(seq, offsets, pos_off, nodes, n_reads, first_read_no, class_vi, flags)
    matcher.sync()


def partitioned_finish(matcher, sums, max_keys, dsums, group=None):
    """merge the per-read statistics (all-reduce) and add up the disjoint per-partition unique counts.
    Returns the global (table, dtable) as numpy arrays, identical on every rank."""
    import numpy as np
    matcher.device_state()  # (the kernels spread their counters over several copies: folded into the arrays below, the stream synchronised)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_keys, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(dsums, op=dist.ReduceOp.SUM, group=group)
    torch.cuda.synchronize(sums.device)
    table, dtable = matcher.finish()
    uniq = torch.from_numpy(np.ascontiguousarray(table[:, 3])).to(sums.device)
    if matcher.config.count_unique:
        dist.all_reduce(uniq, op=dist.ReduceOp.SUM, group=group)
        table[:, 3] = uniq.cpu().numpy()
    return table, dtable
