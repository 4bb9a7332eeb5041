"""world_size-2 gloo test of the read-sharded merge (genestrip_amd/distributed.py), on CPU.

Each rank runs the oracle on its contiguous slice of the reads (global readNo kept), exports the raw
accumulators, merges them with the same function bench.py uses over RCCL, and the merged result must equal
a single-rank run over all reads, bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

N_READS = 6000


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genestrip_amd import synth
    from genestrip_amd.distributed import merge_run_state, shard_bounds
    from oracle import gs_oracle as orc

    db = synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=15000, seed=5)
    seq, off = synth.reads_host(db.genomes, N_READS, read_len=150, seed=31)
    lo, hi = shard_bounds(N_READS, rank, world)
    odb = orc.DB(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    run = orc.MatchRun(odb)
    run.submit(seq, off[lo:hi + 1].copy(), first_read_no=lo)
    table, bitmap = run.export_state()
    # split the oracle table into the library's accumulator layout: additive columns + packed max key
    add_cols = [orc.C_READS, orc.C_READS_KMERS, orc.C_KMERS, orc.C_CONTIGS, orc.C_CONTIG_LEN_SQ_SUM,
                orc.C_READS_1KMER, orc.C_READS_BPS]
    sums = torch.from_numpy(table[:, add_cols].copy())
    mlen, mno = table[:, orc.C_MAX_CONTIG_LEN], table[:, orc.C_MAX_CONTIG_READ_NO]
    keys = np.where(mlen > 0, (mlen << 40) | ((1 << 40) - 1 - np.maximum(mno, 0)), 0).astype(np.int64)
    max_keys = torch.from_numpy(keys)
    bits = torch.from_numpy(bitmap.view(np.int64).copy())
    merge_run_state(sums, max_keys, None, bits)
    merged = table.copy()
    merged[:, add_cols] = sums.numpy()
    k = max_keys.numpy()
    merged[:, orc.C_MAX_CONTIG_LEN] = k >> 40
    merged[:, orc.C_MAX_CONTIG_READ_NO] = np.where(k != 0, (1 << 40) - 1 - (k & ((1 << 40) - 1)), -1)
    run.import_state(merged, bits.numpy().view(np.uint64))
    final, _ = run.finish()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), final)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_merge_equals_single_rank(tmp_path):
    from genestrip_amd import synth
    from oracle import gs_oracle as orc
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    db = synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=15000, seed=5)
    seq, off = synth.reads_host(db.genomes, N_READS, read_len=150, seed=31)
    run = orc.MatchRun(orc.DB(31, db.kmers, db.value_idx, db.n_values, db.parent_vi))
    run.submit(seq, off)
    want, _ = run.finish()
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npy"))
        assert np.array_equal(got, want), np.argwhere(got != want)[:5]


def test_shard_bounds_cover_everything():
    from genestrip_amd.distributed import shard_bounds
    for n in (0, 1, 7, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def _route_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genestrip_amd import distributed as gd
    g = torch.Generator().manual_seed(100 + rank)
    n = 5000 + 37 * rank
    keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, generator=g)
    keys[torch.randint(0, n, (n // 50,), generator=g)] = gd.KEY_INVALID
    keys[torch.randint(0, n, (n // 3,), generator=g)] = gd.KEY_MISS  # ruled out by the gate: never routed
    idx, send, counts = gd.plan_routing(keys, world)
    recv, rc = gd.exchange_all_to_all(send, counts)
    # every key this rank received is one it owns
    assert bool((((recv >> gd.OWNER_SHIFT) % world) == rank).all())
    probe = (recv % 1000).to(torch.int32)  # stand-in for gs_match_probe_keys on the owner
    back, _ = gd.exchange_all_to_all(probe, rc)
    assert int(counts.sum()) == int((keys >= 0).sum())
    nodes = gd.scatter_nodes(back, idx, n, keys)
    want = torch.where(keys == gd.KEY_INVALID, torch.tensor(gd.NODE_INVALID, dtype=torch.int64), keys % 1000)
    want = torch.where(keys == gd.KEY_MISS, torch.tensor(gd.NODE_MISS, dtype=torch.int64), want).to(torch.int32)
    assert torch.equal(nodes, want)
    np.save(os.path.join(out_dir, f"route{rank}.npy"), np.array([int(recv.numel()), int((keys >= 0).sum())]))
    dist.barrier()
    dist.destroy_process_group()


def test_db_partitioned_routing_round_trip_two_ranks(tmp_path):
    """keys -> owner ranks -> (stand-in probe) -> back to the home rank in the original order, over gloo"""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_route_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = sum(np.load(os.path.join(str(tmp_path), f"route{r}.npy")) for r in range(2))
    assert got[0] == got[1] and 5000 < got[0] < 8000  # every routable key arrived somewhere, the others stayed home


def test_position_offsets():
    from genestrip_amd import distributed as gd
    off = torch.tensor([0, 10, 40, 41, 200], dtype=torch.int64)
    assert gd.position_offsets(off, 31).tolist() == [0, 0, 0, 0, 129]
    assert gd.position_offsets(off, 2).tolist() == [0, 9, 38, 38, 196]


def test_bench_refuses_more_gpus_than_visible():
    """bench.py --gpus N without a launcher starts its ranks itself -- and exits non-zero, before any GPU call and without
    printing a line, when fewer than N devices are visible (a SCALE run must never record N copies of the N = 1
    number)"""
    import subprocess
    import sys

    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GS_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(max(2, n))], capture_output=True, text=True,
                       timeout=300, env=env, cwd=root)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert "device(s) visible" in r.stderr and not [x for x in r.stdout.splitlines() if x.startswith("{")]


def test_bench_parent_never_maps_the_hip_runtime(tmp_path):
    """the process that starts the ranks of `bench.py --gpus N` must not have initialised the GPU runtime when it
    forks + execs the launcher.  It does not even import torch: the devices are counted from the KFD topology.  The parent writes
    what it saw into GS_BENCH_LAUNCH_HOOK just before it decides."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hook = str(tmp_path / "hook.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GS_BENCH_BACKEND")}
    env["GS_BENCH_LAUNCH_HOOK"] = hook
    env["ROCR_VISIBLE_DEVICES"] = ""  # (no device for the ranks, wherever this runs: the refusal path)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    seen = json.load(open(hook))
    assert r.returncode == 3 and seen["visible"] == 0, (r.returncode, seen, r.stderr[-300:])
    assert seen["hip_mapped"] is False and seen["torch_imported"] is False, seen


def test_partitioned_batch_refuses_positions_beyond_the_routing_index():
    """a routed key carries its batch position as 32 bits; a batch with more positions must be refused, not truncated"""
    from genestrip_amd import distributed as gd
    assert gd.check_batch_positions(0) == 0 and gd.check_batch_positions(gd.MAX_BATCH_POSITIONS) == gd.MAX_BATCH_POSITIONS
    with pytest.raises(ValueError):
        gd.check_batch_positions(gd.MAX_BATCH_POSITIONS + 1)
    with pytest.raises(ValueError):
        gd.check_batch_positions(1 << 33)
