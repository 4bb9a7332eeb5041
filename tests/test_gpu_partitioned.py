"""DB-partitioned match (SURVEY 8e / BASELINE configs[4]) on one GPU: the split pipeline
encode -> (all-to-all) -> probe on the owner -> (all-to-all) -> reduce must give the oracle's table.
The W-rank exchange is emulated in-process with W partitioned stores on the same GPU, using the same routing
helpers the torch.distributed path uses; the collective path itself is exercised with a 1-rank RCCL group."""
import os

import numpy as np
import pytest
import torch

import genestrip_amd as ga
from genestrip_amd import distributed as gd
from genestrip_amd import synth
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu
ADD = [0, 1, 2, 3, 4, 5, 7, 8]  # additive columns (unique counts of the partitions are disjoint)


@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)


def _reads(sdb, n):
    seq, off = synth.reads_host(sdb.genomes, n, read_len=150, seed=61)
    rng = np.random.default_rng(2)
    seq = seq.copy()
    for r in rng.choice(n, n // 40, replace=False):
        seq[int(off[r]) + int(rng.integers(0, 150))] = ord("N")
    # ragged tail: short, long and empty reads
    extra = [b"", b"ACGT", bytes(sdb.genomes[1][:31]), bytes(sdb.genomes[2][100:900]), bytes(sdb.genomes[0][:40]) + b"N" * 3]
    # reads from elsewhere: almost all of their k-mers are ruled out by the minimizer gate and never routed
    extra += [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(ln))) for ln in rng.integers(31, 400, 300)]
    eseq, eoff = orc.pack_reads(extra)
    seq = np.concatenate([seq, eseq])
    off = np.concatenate([off, off[-1] + eoff[1:]])
    return seq, off


def _oracle(sdb, seq, off, **cfg):
    run = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi), **cfg)
    cv, fl = run.submit(seq, off)
    t, _ = run.finish()
    return t, cv, fl


def _merge_tables(tables):
    out = np.zeros_like(tables[0])
    for c in ADD:
        out[:, c] = sum(t[:, c] for t in tables)
    for v in range(out.shape[0]):
        best = (0, -1)
        for t in tables:
            if t[v, 6] > best[0] or (t[v, 6] == best[0] and t[v, 6] > 0 and t[v, 9] < best[1]):
                best = (int(t[v, 6]), int(t[v, 9]))
        out[v, 6], out[v, 9] = best
    return out


@pytest.mark.parametrize("device_routing", [True, False])
@pytest.mark.parametrize("world", [1, 2, 3])
def test_partitioned_pipeline_emulated(sdb, world, device_routing):
    seq, off = _reads(sdb, 6000)
    n = len(off) - 1
    want, wcv, wfl = _oracle(sdb, seq, off)
    dev = torch.device("cuda")
    stores = [ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, n_parts=world, part=p, partition=True)
              for p in range(world)]
    assert sum(s.info.n_stored for s in stores) == ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values,
                                                                      sdb.parent_vi).info.n_stored
    ms = [ga.FastqKMerMatcher(s) for s in stores]
    # read shards per rank (global read numbers kept)
    bounds = [gd.shard_bounds(n, r, world) for r in range(world)]
    shard = []
    for lo, hi in bounds:
        dseq = torch.from_numpy(seq[int(off[lo]):int(off[hi])].copy()).to(dev) if hi > lo else torch.zeros(1, dtype=torch.uint8, device=dev)
        doff = torch.from_numpy((off[lo:hi + 1] - off[lo]).astype(np.int64)).to(dev)
        shard.append((dseq, doff, hi - lo, lo))
    plans = []
    n_gated = 0
    for r, (dseq, doff, nr, lo) in enumerate(shard):
        pos_off = gd.position_offsets(doff, 31)
        nk = int(pos_off[-1].item())
        keys = torch.empty(max(nk, 1), dtype=torch.int64, device=dev)
        ms[r].encode(dseq, doff, pos_off, keys, nr)
        ms[r].sync()
        n_gated += int((keys[:nk] == gd.KEY_MISS).sum().item())
        if device_routing:  # gs_route_keys: counting sort on the device
            send = torch.empty(max(nk, 1), dtype=torch.int64, device=dev)
            idx = torch.empty(max(nk, 1), dtype=torch.int32, device=dev)
            early = torch.full((max(nk, 1),), -9, dtype=torch.int32, device=dev) if r % 2 == 0 else None
            counts = np.array(ms[r].route_keys(keys, nk, world, send, idx, early), dtype=np.int64)
            send = send[:int(counts.sum())]
            assert bool((((send >> gd.OWNER_SHIFT) % world).cpu() == torch.repeat_interleave(
                torch.arange(world), torch.from_numpy(counts))).all())
        else:               # the same step with torch ops
            idx, send, counts = gd.plan_routing(keys[:nk], world)
            counts = counts.cpu().numpy()
        plans.append((pos_off, nk, idx, send, counts, keys, early if device_routing else None))
    assert n_gated > 30000  # the foreign reads' k-mers (about 55 k) stop at the gate of the encoding rank
    # all-to-all #1: keys to their owners
    starts = [np.concatenate([[0], np.cumsum(p[4])]) for p in plans]
    node_back = [[None] * world for _ in range(world)]
    for j in range(world):
        parts = [plans[i][3][starts[i][j]:starts[i][j + 1]] for i in range(world)]
        recv = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=dev)
        nodes = torch.empty(max(recv.numel(), 1), dtype=torch.int32, device=dev)
        ms[j].probe_keys(recv, nodes, recv.numel())
        ms[j].sync()
        o = 0
        for i in range(world):  # all-to-all #2: nodes back to the home ranks
            c = int(plans[i][4][j])
            node_back[i][j] = nodes[o:o + c]
            o += c
    tables, cvs, fls = [], [], []
    for r, (dseq, doff, nr, lo) in enumerate(shard):
        pos_off, nk, idx, _, _, keys, early = plans[r]
        back = torch.cat(node_back[r])
        if device_routing:
            # both ways of filling in the unrouted positions: by gs_route_keys already, or by gs_unroute_nodes
            nodes = early if early is not None else torch.empty(max(nk, 1), dtype=torch.int32, device=dev)
            ms[r].unroute_nodes(None if early is not None else keys, idx, back, back.numel(), nodes, nk)
        else:
            nodes = gd.scatter_nodes(back, idx, max(nk, 1), keys)
        cv = torch.full((max(nr, 1),), -1, dtype=torch.int32, device=dev)
        fl = torch.zeros(max(nr, 1), dtype=torch.uint8, device=dev)
        ms[r].reduce(dseq, doff, pos_off, nodes, nr, first_read_no=lo, class_vi=cv, flags=fl)
        t, _ = ms[r].finish()
        tables.append(t)
        cvs.append(cv[:nr].cpu().numpy())
        fls.append(fl[:nr].cpu().numpy())
    got = _merge_tables(tables)
    assert np.array_equal(got, want), np.argwhere(got != want)[:8]
    assert np.array_equal(np.concatenate(cvs), wcv) and np.array_equal(np.concatenate(fls), wfl)
    for m in ms:
        m.close()
    for s in stores:
        s.close()


def test_partitioned_collective_path_single_rank(sdb):
    """the torch.distributed code path (all_to_all_single, all_reduce over RCCL) with a 1-rank group"""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        seq, off = _reads(sdb, 3000)
        n = len(off) - 1
        want, wcv, wfl = _oracle(sdb, seq, off)
        store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, n_parts=1, part=0, partition=True)
        m = ga.FastqKMerMatcher(store)
        dseq = torch.from_numpy(seq).to(dev)
        doff = torch.from_numpy(off.astype(np.int64)).to(dev)
        cv = torch.full((n,), -1, dtype=torch.int32, device=dev)
        fl = torch.zeros(n, dtype=torch.uint8, device=dev)
        gd.partitioned_match_batch(m, 31, dseq, doff, n, 0, class_vi=cv, flags=fl)
        st = m.device_state()

        class V:
            def __init__(self, ptr, cnt, t):
                self.__cuda_array_interface__ = {"data": (ptr, False), "shape": (cnt,), "typestr": t, "version": 2}
        sums = torch.as_tensor(V(st["sums"], sdb.n_values * ga.N_SUMS, "<i8"), device=dev)
        mx = torch.as_tensor(V(st["max_keys"], sdb.n_values, "<i8"), device=dev)
        ds = torch.as_tensor(V(st["dsums"], sdb.n_values * ga.N_DCOLS, "<f8"), device=dev)
        table, _ = gd.partitioned_finish(m, sums, mx, ds)
        assert np.array_equal(table, want)
        assert np.array_equal(cv.cpu().numpy(), wcv) and np.array_equal(fl.cpu().numpy(), wfl)
        m.close()
        store.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 3, 8])
def test_fused_encode_route_equals_encode_plus_route(sdb, world):
    """gs_match_encode_route against gs_match_encode + gs_route_keys: the same keys with the same positions in every
    owner's region (in any order), sentinels in the unused slots, the same nodes for the positions that are not routed"""
    seq, off = _reads(sdb, 5000)
    n = len(off) - 1
    dev = torch.device("cuda")
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, n_parts=world, part=0, partition=True)
    m = ga.FastqKMerMatcher(store)
    dseq = torch.from_numpy(seq).to(dev)
    doff = torch.from_numpy(off.astype(np.int64)).to(dev)
    pos_off = gd.position_offsets(doff, 31)
    nk = int(pos_off[-1].item())
    keys = torch.empty(nk, dtype=torch.int64, device=dev)
    m.encode(dseq, doff, pos_off, keys, n)
    m.sync()
    send = torch.empty(nk, dtype=torch.int64, device=dev)
    idx = torch.empty(nk, dtype=torch.int32, device=dev)
    nodes0 = torch.full((nk,), -9, dtype=torch.int32, device=dev)
    counts0 = m.route_keys(keys, nk, world, send, idx, nodes0)
    cap = ((nk // world) * 2 + 8192 * gd.ROUTE_CHUNK + gd.ROUTE_CHUNK - 1) // gd.ROUTE_CHUNK * gd.ROUTE_CHUNK
    sk = torch.empty(world * cap, dtype=torch.int64, device=dev)
    si = torch.empty(world * cap, dtype=torch.int32, device=dev)
    nodes1 = torch.full((nk,), -9, dtype=torch.int32, device=dev)
    counts1, over = m.encode_route(dseq, doff, pos_off, n, world, cap, sk, si, nodes1)
    assert not over
    start = 0
    for o in range(world):
        assert counts1[o] % gd.ROUTE_CHUNK == 0 and counts1[o] >= counts0[o]
        rk, ri = sk[o * cap:o * cap + counts1[o]].cpu().numpy(), si[o * cap:o * cap + counts1[o]].cpu().numpy()
        used = ri != -1
        assert np.all(rk[~used] == -1)  # sentinels
        want_k, want_i = send[start:start + counts0[o]].cpu().numpy(), idx[start:start + counts0[o]].cpu().numpy()
        a, b = np.argsort(ri[used], kind="stable"), np.argsort(want_i, kind="stable")
        assert np.array_equal(ri[used][a], want_i[b]) and np.array_equal(rk[used][a], want_k[b])
        start += counts0[o]
    routed = (keys >= 0).cpu().numpy()
    assert np.array_equal(nodes1.cpu().numpy()[~routed], nodes0.cpu().numpy()[~routed])
    # a region that is too small is reported, not overrun
    small = gd.ROUTE_CHUNK * 4
    sk2 = torch.full((world * small + 1,), 12345, dtype=torch.int64, device=dev)
    si2 = torch.empty(world * small, dtype=torch.int32, device=dev)
    c2, over2 = m.encode_route(dseq, doff, pos_off, n, world, small, sk2, si2, nodes1)
    assert over2 and all(c <= small for c in c2) and int(sk2[-1].item()) == 12345
    m.close()
    store.close()


def test_encode_route_refuses_a_batch_whose_positions_overflow_the_routing_index(sdb):
    """gs_match_encode_route stores a routed key's batch position as 32 bits (~0 = unused slot); pos_off[n_reads] >=
    2^32 - 1 must come back as GS_E_INVALID before anything is launched (gs_route_keys refuses the same)"""
    dev = torch.device("cuda", 0)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, n_parts=1, part=0, partition=True)
    m = ga.FastqKMerMatcher(store)
    seq, off = _reads(sdb, 4)
    dseq = torch.from_numpy(seq).to(dev)
    doff = torch.from_numpy(off.astype(np.int64)).to(dev)
    for total, ok in ((480, True), (0xfffffffe, True), (0xffffffff, False), (1 << 33, False)):
        pos = torch.tensor([0, 120, 240, 360, total], dtype=torch.int64, device=dev)
        keys = torch.empty(2048, dtype=torch.int64, device=dev)
        idx = torch.empty(2048, dtype=torch.int32, device=dev)
        nodes = torch.empty(600, dtype=torch.int32, device=dev)
        if ok and total > 480:
            continue  # (a legal size this large would need 16 GB of nodes; the boundary itself is checked from the refusing side)
        if ok:
            m.encode_route(dseq, doff, pos, 4, 1, 2048, keys, idx, nodes)
        else:
            with pytest.raises(ga.GsError) as e:
                m.encode_route(dseq, doff, pos, 4, 1, 2048, keys, idx, nodes)
            assert e.value.code == -1
    m.close()
    store.close()


def test_partitioned_batch_fused_unfused_and_overflow_fallback(sdb):
    """partitioned_match_batch under a 1-rank RCCL group: the fused route, the explicit unfused route and the fallback
    when a region is too small all end in the oracle's table"""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        seq, off = _reads(sdb, 3000)
        n = len(off) - 1
        want, wcv, wfl = _oracle(sdb, seq, off)
        store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, n_parts=1, part=0, partition=True)
        dseq = torch.from_numpy(seq).to(dev)
        doff = torch.from_numpy(off.astype(np.int64)).to(dev)
        for how in ("fused", "unfused", "overflow"):
            m = ga.FastqKMerMatcher(store)
            cv = torch.full((n,), -1, dtype=torch.int32, device=dev)
            fl = torch.zeros(n, dtype=torch.uint8, device=dev)
            fell_back = gd.ROUTE_OVERFLOWS[0]
            if how == "unfused":
                gd.partitioned_match_batch_unfused(m, 31, dseq, doff, n, 0, class_vi=cv, flags=fl)
            else:
                gd.partitioned_match_batch(m, 31, dseq, doff, n, 0, class_vi=cv, flags=fl, cap=None if how == "fused" else gd.ROUTE_CHUNK)
            # the default region size comes from the library's own launch geometry: it never overflows; a region of one chunk does
            assert gd.ROUTE_OVERFLOWS[0] - fell_back == (1 if how == "overflow" else 0), how
            waves, chunk = m.route_geometry(n)
            assert chunk == gd.ROUTE_CHUNK and waves % 4 == 0 and 4 <= waves <= torch.cuda.get_device_properties(dev).multi_processor_count * 32
            table, _ = m.finish()
            assert np.array_equal(table, want), how
            assert np.array_equal(cv.cpu().numpy(), wcv) and np.array_equal(fl.cpu().numpy(), wfl), how
            m.close()
        store.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["partitioned", "sharded", "striped"])
def test_two_rank_bench_flow_over_gloo(mode):
    """bench.py's multi-rank control flow (barriers, routing all-to-alls or state merge, rank-0-only legs) with two
    processes on this one GPU (striped: each rank owns one stripe of the record table and reads the other through a HIP IPC
    handle -- real cross-process peer mapping, only not across xGMI); the collectives run over gloo on host copies (GS_BENCH_BACKEND=gloo, a rehearsal mode:
    the numbers mean nothing).  Both ranks must end with the same merged table, bit-exact against the oracle."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GS_BENCH_BACKEND="gloo")
    port = 29800 + (os.getpid() % 1500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--reads", "300000", "--check-reads", "50000", "--cpu-seconds", "0", "--mode", mode]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["parity"]["bit_exact"] is True
    assert out["parity"]["merged_table_identical_on_all_ranks"] is True
