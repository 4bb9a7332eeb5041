"""the oracle's restatement of RadixKMerStore (C/store/RadixKMerStore.java:369-412 getLong, :319-364
putLong, :632-671 optimize, :714-729 visit, :160-164 value cap) against its KMerSortedArray restatement: the two layouts
must answer every lookup alike and give identical match tables.  CPU only."""
import numpy as np
import pytest

from genestrip_amd import synth
from oracle import gs_oracle as orc


def _wide_tree(n_values, fan=300):
    """root 0, `fan` inner nodes, the other values leaves spread under them (value index == pre-order is not needed)"""
    parent = np.empty(n_values, dtype=np.int32)
    parent[0] = -1
    parent[1:fan + 1] = 0
    parent[fan + 1:] = 1 + (np.arange(n_values - fan - 1) % fan)
    return parent


def test_radix_value_cap_follows_the_reference():
    # maxValuesForRadix: valueBits = min(30, 64 - (62 - radixBits)) (:160-164); [16, 30] only (:166-171)
    assert orc.radix_max_values(16) == 1 << 18 and orc.radix_max_values(17) == 1 << 19
    assert orc.radix_max_values(28) == 1 << 30 and orc.radix_max_values(30) == 1 << 30
    assert orc.radix_max_values(15) == -1 and orc.radix_max_values(31) == -1
    with pytest.raises(ValueError):
        orc.DB(31, [5, 9], [0, 1], (1 << 18) + 1, None, radix_bits=16)


@pytest.mark.parametrize("k,radix_bits,gate", [(31, 17, False), (31, 16, True), (21, 20, False), (12, 17, True)])
def test_radix_getlong_equals_sorted_getlong(k, radix_bits, gate):
    rng = np.random.default_rng(k * 100 + radix_bits)
    hi = 1 << (2 * k)
    keys = np.unique(rng.integers(0, hi, 60000, dtype=np.int64))
    # clusters inside one radix bucket: same low bits, different remaining bits (the binary search has work to do)
    base = keys[:50] & ((1 << radix_bits) - 1)
    extra = (rng.integers(0, hi >> radix_bits, (50, 40), dtype=np.int64) << radix_bits) | base[:, None]
    keys = np.unique(np.concatenate([keys, extra.ravel()]))
    vals = rng.integers(0, 1000, len(keys)).astype(np.int32)
    sorted_db = orc.DB(k, keys, vals, 1000, None, gate)
    perm = rng.permutation(len(keys))  # putLong order is arbitrary
    radix_db = orc.DB(k, keys[perm], vals[perm], 1000, None, gate, radix_bits=radix_bits)
    vk, vv = radix_db.visit()
    # visit order: by bucket (low radix bits), inside a bucket by the remaining bits; positions are dense
    order = np.lexsort((vk >> radix_bits, vk & ((1 << radix_bits) - 1)))
    assert np.array_equal(order, np.arange(len(vk)))
    assert np.array_equal(np.sort(vk), keys) and np.array_equal(vv[np.argsort(vk)], vals)
    probes = np.concatenate([keys[::7], rng.integers(0, hi, 20000, dtype=np.int64), keys[:200] ^ (1 << radix_bits)])
    pos_of = {int(kk): i for i, kk in enumerate(vk)}
    for q in probes.tolist():
        a, apos = sorted_db.get(q)
        b, bpos = radix_db.get(q)
        assert a == b
        if b >= 0:
            assert keys[apos] == q and bpos == pos_of[q]


@pytest.mark.parametrize("n_values,radix_bits", [(2500, 17), (70000, 17), (70000, 16)])
def test_radix_and_sorted_layout_give_identical_tables(n_values, radix_bits):
    """> 65 535 values is legal only for the radix store (KMerSortedArray caps at 65 535, :56): the restated sorted
    array has int32 value indices and serves as the layout-independent expectation"""
    db = synth.SynthDB(k=31, genera=2, species_per_genus=3, genome_len=30000, seed=5)
    rng = np.random.default_rng(n_values)
    parent = _wide_tree(n_values)
    vals = rng.integers(0, n_values, len(db.kmers)).astype(np.int32)
    seq, off = synth.reads_host(db.genomes, 4000, read_len=150, seed=9)
    res = []
    for rb in (0, radix_bits):
        if rb:
            perm = rng.permutation(len(db.kmers))
            odb = orc.DB(31, db.kmers[perm], vals[perm], n_values, parent, True, radix_bits=rb)
        else:
            odb = orc.DB(31, db.kmers, vals, n_values, parent, True)
        run = orc.MatchRun(odb, max_kmer_res_counts=3)
        cv, fl = run.submit(seq, off, threads=4)
        mc = run.max_counts()
        t, d = run.finish()
        res.append((t, cv, fl, mc))
    assert res[0][0][:, orc.C_KMERS].sum() > 100000
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
