"""Kraken-style per-read segments from the GPU (gs_match_segments) against the oracle's restatement of
printKrakenStyleOut, plus the reference's golden line for dengue1 (R/projects/dengue1/test.out)."""
import os

import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import synth
from conftest import GOLDEN
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu


def _gpu_segments(m, reads):
    seq, off = orc.pack_reads(reads)
    seg_off, codes, starts, lens = m.segments(seq, off)
    return [list(zip(codes[int(a):int(b)].tolist(), lens[int(a):int(b)].tolist())) for a, b in zip(seg_off[:-1], seg_off[1:])]


def test_dengue_golden_line():
    lines = open(os.path.join(GOLDEN, "dengue1", "dengue1.fasta")).read().split("\n")
    genome = "".join(l.strip() for l in lines if not l.startswith(">")).upper()
    keys = np.unique(orc.canonical_kmers(genome, 31))
    store = ga.DeviceKMerStore(31, keys, np.zeros(len(keys), np.int32), 1, np.array([-1], np.int32))
    m = ga.FastqKMerMatcher(store)
    rd = orc.parse_fastq(open(os.path.join(GOLDEN, "dengue1", "test.fastq"), "rb").read(), k=31)
    read = rd["seq"].tobytes()
    segs = _gpu_segments(m, [read])[0]
    cv, fl = m.match_reads(rd["seq"], rd["seq_off"])
    names = {0: "1", -1: "0", -2: "A"}
    desc = rd["desc"].tobytes()[1:].split(b" ")[0].decode()
    line = ("C" if cv[0] >= 0 else "U") + f"\t{desc}\t" + ("1" if cv[0] == 0 else "0") + f"\t{len(read)}\t" + \
        " ".join(f"{names[c]}:{n}" for c, n in segs)
    assert line == open(os.path.join(GOLDEN, "dengue1", "test.out")).read().rstrip("\n")
    m.close()
    store.close()


def test_segments_match_oracle_on_ragged_reads():
    sdb = synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)
    rng = np.random.default_rng(23)
    g0 = sdb.genomes
    reads = [b"", b"ACGT", bytes(g0[0][:31]), b"N" * 50, bytes(g0[1][:40]) + b"N" + bytes(g0[1][41:90])]
    for _ in range(500):
        parts = []
        for _ in range(int(rng.integers(1, 5))):
            s = int(rng.integers(0, g0.shape[0]))
            L = int(rng.integers(20, 400))
            p = int(rng.integers(0, g0.shape[1] - L))
            parts.append(g0[s][p:p + L].tobytes())
        r = bytearray(b"".join(parts))
        for _ in range(int(rng.integers(0, 5))):
            r[int(rng.integers(0, len(r)))] = ord("N") if rng.random() < 0.5 else rng.choice(list(b"ACGTacgt"))
        reads.append(bytes(r))
    reads.append(bytes(g0[3][:3000]))
    odb = orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    got = _gpu_segments(m, reads)
    for i, r in enumerate(reads):
        want = odb.segments(r)
        assert got[i] == want, (i, len(r), got[i][:6], want[:6])
        if len(r) >= 31:
            assert sum(n for _, n in got[i]) == len(r) - 30
    m.close()
    store.close()
