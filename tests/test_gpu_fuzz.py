"""Randomised parity scenarios for the match path: k, taxonomy shape, value count, thresholds, read lengths and -- on
purpose -- low-complexity sequence (homopolymers, short tandem repeats), where many k-mers share a minimizer, the same
15-mer occurs twice inside a k-mer and a k-mer can equal its own reverse complement.  Everything is compared with the
oracle bit for bit.  Needs an MI355X: run with -m gpu."""
import os

import numpy as np
import pytest

import genestrip_amd as ga
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu


def _genome(rng, n):
    """random sequence with stretches of low complexity"""
    out = bytearray()
    while len(out) < n:
        kind = rng.integers(0, 6)
        if kind <= 2:
            out += bytes(rng.choice(list(b"ACGT"), int(rng.integers(50, 400))).tolist())
        elif kind == 3:
            out += bytes([rng.choice(list(b"ACGT"))]) * int(rng.integers(20, 90))           # homopolymer
        elif kind == 4:
            unit = bytes(rng.choice(list(b"ACGT"), int(rng.integers(2, 7))).tolist())
            out += unit * int(rng.integers(8, 40))                                          # tandem repeat
        else:
            half = bytes(rng.choice(list(b"ACGT"), int(rng.integers(10, 40))).tolist())
            comp = bytes({65: 84, 67: 71, 71: 67, 84: 65}[c] for c in reversed(half))
            out += half + comp                                                              # reverse-complement palindrome
    return bytes(out[:n])


def _tree(rng, n_values):
    """random rooted tree over value indices 0..n-1 (0 = root), sometimes a long chain"""
    parent = np.full(n_values, -1, dtype=np.int32)
    chain = rng.random() < 0.3
    for v in range(1, n_values):
        parent[v] = v - 1 if chain and v < 40 else int(rng.integers(0, v))
    return parent


# GS_FUZZ_EXTRA=n: n more scenarios (seeds 24 ..), which also draw maxClassificationPaths = 128 (a campaign, not part of the suite)
@pytest.mark.parametrize("seed", range(24 + int(os.environ.get("GS_FUZZ_EXTRA", "0"))))
def test_random_scenario(seed):
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([19, 21, 25, 31, 31, 31, 12, 17]))
    n_values = int(rng.choice([3, 9, 40, 300, 2500]))
    parent = _tree(rng, n_values)
    n_gen = min(n_values - 1, int(rng.integers(2, 9)))
    leaves = rng.choice(np.arange(1, n_values), n_gen, replace=False)
    genomes = [_genome(rng, int(rng.integers(1500, 6000))) for _ in range(n_gen)]
    odb_tmp = None
    d = {}
    for vi, g in zip(leaves.tolist(), genomes):
        for x in orc.canonical_kmers(g, k).tolist():
            if x in d and d[x] != vi:
                # shared k-mer: the lowest common ancestor, as DBGoal assigns it
                if odb_tmp is None:
                    odb_tmp = orc.DB(k, np.array([0], dtype=np.int64), np.array([0], dtype=np.int32), n_values, parent)
                d[x] = odb_tmp.lca(d[x], vi)
            else:
                d[x] = vi
    keys = np.array(sorted(d), dtype=np.int64)
    vidx = np.array([d[x] for x in keys.tolist()], dtype=np.int32)
    cfg = dict(classify=bool(rng.random() < 0.85), count_unique=bool(rng.random() < 0.8),
               max_paths=int(rng.choice([1, 2, 4, 10, 64] + ([128] if seed >= 24 else []))), threshold=int(rng.choice([1, 1, 2, 5, 30])),
               max_read_tax_err=float(rng.choice([-1.0, 0.0, 0.2, 3.0])),
               max_read_class_err=float(rng.choice([-1.0, 0.1, 0.6, 25.0])))
    reads = []
    for _ in range(900):
        g = genomes[int(rng.integers(0, n_gen))]
        L = int(rng.choice([int(rng.integers(0, 60)), int(rng.integers(60, 300)), int(rng.integers(300, 1500))], p=[0.15, 0.7, 0.15]))
        L = min(L, len(g))
        p = int(rng.integers(0, len(g) - L + 1))
        r = bytearray(g[p:p + L])
        if rng.random() < 0.5:  # the other strand
            r = bytearray({65: 84, 67: 71, 71: 67, 84: 65}[c] for c in reversed(r))
        for _ in range(int(rng.integers(0, 5))):
            if L:
                r[int(rng.integers(0, L))] = int(rng.choice(list(b"ACGTNacgt-")))
        reads.append(bytes(r))
    seq, off = orc.pack_reads(reads)
    first = int(rng.integers(0, 1 << 30))
    orun = orc.MatchRun(orc.DB(k, keys, vidx, n_values, parent), **cfg)
    ocv, ofl = orun.submit(seq, off, first)
    ot, _ = orun.finish()
    assert int((ofl & orc.F_FOUND != 0).sum()) > 100  # the scenario is not vacuous
    store = ga.DeviceKMerStore(k, keys, vidx, n_values, parent)
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(**cfg))
    gcv, gfl = m.match_reads(seq, off, first)
    gt, _ = m.finish()
    # the same reads again as raw FASTQ text (device-side record scan), into a fresh state
    m.reset()
    text = b"".join(b"@x\n" + r + b"\n+\n" + b"#" * len(r) + b"\n" for r in reads)
    cv2 = np.zeros(len(reads), dtype=np.int32)
    fl2 = np.zeros(len(reads), dtype=np.uint8)
    m.submit_text(text, first_read_no=first, class_vi=cv2, flags=fl2)
    m.sync()
    assert m.text_status()[0] == -1
    gt2, _ = m.finish()
    m.close()
    store.close()
    what = (seed, k, n_values, cfg)
    assert np.array_equal(gt, ot), (what, np.argwhere(gt != ot)[:6])
    assert np.array_equal(gcv, ocv) and np.array_equal(gfl, ofl), what
    assert np.array_equal(gt2, ot) and np.array_equal(cv2, ocv) and np.array_equal(fl2, ofl), what
