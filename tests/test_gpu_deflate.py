"""The output side on the device (genestrip_amd/csrc/gs_deflate_dev.hip): block-gzip members written by the GPU must inflate -- under
zlib, as java.util.zip.GZIPInputStream would read them (B/io/StreamProvider.java:92-100), AND under this library's device inflater --
to exactly the text they were made from; and the records gathered on the device must be the bytes ReadEntry.write produces
(C/fastq/AbstractFastqReader.java:570-584) for the reads the filter accepted / matchRead returned (C/bloom/FastqBloomFilter.java:92-105,
C/match/FastqKMerMatcher.java:304-307)."""
import gzip
import zlib

import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import synth
from oracle import gs_oracle as orc
from test_deflate_cpu import CASES, _fastq, n_members

pytestmark = pytest.mark.gpu


def _check_members(comp, data):
    members, reached = ga.bgzf_members(comp)
    assert reached == len(comp) and len(members) == n_members(len(data))
    assert gzip.decompress(comp + ga.BGZF_EOF) == data
    at = 0
    for po, pl, isz, crc in members:
        text = zlib.decompress(comp[po:po + pl], -15)
        assert len(text) == isz and zlib.crc32(text) == crc and text == data[at:at + isz]
        at += isz
    return members


@pytest.mark.parametrize("name", list(CASES))
def test_device_members_inflate_under_zlib_and_on_the_device(name):
    data = CASES[name]
    comp = ga.deflate_device(data).tobytes()
    members = _check_members(comp, data)
    text, status = ga.inflate_members(comp, members)
    assert not status.any() and text.tobytes() == data


def test_whole_sized_pieces(monkeypatch):
    """a long text is cut into 63 KiB members (here forced: GS_DEFLATE_PIECE)"""
    monkeypatch.setenv("GS_DEFLATE_PIECE", "64512")
    data = _fastq(3000, seed=33) + b"@tail\nACGT"
    comp = ga.deflate_device(data).tobytes()
    members, reached = ga.bgzf_members(comp)
    assert reached == len(comp) and len(members) == (len(data) + 64511) // 64512 and members[0][2] == 64512
    assert gzip.decompress(comp + ga.BGZF_EOF) == data
    text, status = ga.inflate_members(comp, members)
    assert not status.any() and text.tobytes() == data


def test_empty_text():
    assert len(ga.deflate_device(b"")) == 0


def test_large_fastq_many_pieces_and_ratio():
    data = _fastq(40000, seed=21)  # 12.9 MB, 200 members
    comp = ga.deflate_device(data).tobytes()
    members = _check_members(comp, data)
    text, status = ga.inflate_members(comp, members)
    assert not status.any() and text.tobytes() == data
    assert len(comp) < len(zlib.compress(data[:2_000_000], 1)) * len(data) / 2_000_000  # bases as two-bit literals beat zlib's level 1


def test_random_texts_and_alphabets():
    rng = np.random.default_rng(5)
    for trial in range(40):
        n = int(rng.integers(1, 200000))
        kind = trial % 5
        if kind == 0:
            data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        elif kind == 1:
            data = bytes(rng.choice(np.frombuffer(b"ACGT\n", dtype=np.uint8), n))
        elif kind == 2:  # runs of random length
            parts = []
            while sum(map(len, parts)) < n:
                parts.append(bytes([int(rng.integers(33, 127))]) * int(rng.integers(1, 700)))
            data = b"".join(parts)[:n]
        elif kind == 3:  # far copies
            blk = bytes(rng.integers(65, 91, 3000, dtype=np.uint8))
            data = (blk + bytes(rng.integers(0, 256, 5000, dtype=np.uint8))) * (n // 8000 + 1)
            data = data[:n]
        else:
            data = _fastq(n // 330 + 1, seed=trial, probs=bool(trial & 8))[:n]
        comp = ga.deflate_device(data).tobytes()
        assert gzip.decompress(comp + ga.BGZF_EOF) == data, (trial, kind, n)


@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)


def _records(seq, off, probs_line):
    """four-line FASTQ with descriptors of varying length, a '+' line that repeats the name now and then, qualities longer than the
    read now and then (legal: the reference takes the line whole)"""
    rng = np.random.default_rng(17)
    recs = []
    for i in range(len(off) - 1):
        s = seq[int(off[i]):int(off[i + 1])].tobytes()
        q = bytes(rng.integers(35, 74, len(s) + (3 if i % 11 == 0 else 0), dtype=np.uint8)) if probs_line else b"I" * len(s)
        recs.append((b"@r%d%s" % (i, b" x" * (i % 4)), s, b"+" + (b"r%d" % i if i % 5 == 0 else b""), q))
    return recs


def _rewritten(recs, keep, with_probs):
    out = []
    for (d, s, _, q), k in zip(recs, keep):
        if k:
            out.append(d + b"\n" + s + b"\n+\n" + (q if with_probs else b"~" * len(s)) + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("with_probs", [False, True])
def test_filter_records_gathered_on_the_device(sdb, with_probs):
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:5])]
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys)
    gb = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
    seq, off = synth.reads_host(sdb.genomes, 9000, read_len=150, seed=5)
    recs = _records(seq, off, True)
    text = b"".join(d + b"\n" + s + b"\n" + p + b"\n" + q + b"\n" for d, s, p, q in recs)
    flt = ga.FastqBloomFilter(31, gb, 1, 0.2)
    acc = np.zeros(len(recs), dtype=np.uint8)
    flt.submit_text(text, acc)
    failed, _, _ = flt.text_status()
    assert failed < 0
    want = ob.filter_batch(31, 1, 0.2, seq, off)
    assert np.array_equal(acc, want) and 0 < acc.sum() < len(acc)
    got, n = flt.compact_text(1, with_probs, slot=0)
    assert n == int(acc.sum()) and got.tobytes() == _rewritten(recs, acc, with_probs)
    rest, n0 = flt.compact_text(0, with_probs, slot=1)
    assert n0 == len(acc) - int(acc.sum()) and rest.tobytes() == _rewritten(recs, 1 - acc, with_probs)
    # the same chunk once more into the other slot: the first result is still there on the device (two chunks under way)
    again, _ = flt.compact_text(1, with_probs, slot=1)
    assert again.tobytes() == got.tobytes()
    gb.close()


def test_match_records_gathered_on_the_device(sdb):
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    seq, off = synth.reads_host(sdb.genomes, 6000, read_len=150, seed=8)
    recs = _records(seq, off, False)
    text = b"".join(d + b"\n" + s + b"\n" + p + b"\n" + q + b"\n" for d, s, p, q in recs)
    cls = np.zeros(len(recs), dtype=np.int32)
    fl = np.zeros(len(recs), dtype=np.uint8)
    m.submit_text(text, class_vi=cls, flags=fl)
    m.sync()
    orun = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi))
    ocv, ofl = orun.submit(seq, off)
    assert np.array_equal(fl, ofl)
    keep = (ofl & orc.F_RETURNED) != 0
    got, n = m.compact_text(False)
    assert n == int(keep.sum()) and 0 < n < len(recs) and got.tobytes() == _rewritten(recs, keep, False)
    m.close()
    store.close()


def test_gather_needs_a_four_line_chunk_with_flags(sdb):
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    with pytest.raises(ga.GsError):
        m.compact_text()
    m.submit_text(b"@a\nACGT\n+\nIIII\n")  # no per-read outputs asked for
    with pytest.raises(ga.GsError):
        m.compact_text()
    m.close()
    store.close()


def test_chunks_compressed_together():
    """gs_deflater_append / _flush: chunks wait on the device and leave as the members of ONE call; the members inflate to the chunks
    in order, and nothing waits afterwards"""
    import torch
    d = ga.DeviceDeflater()
    parts = [_fastq(700, seed=s) for s in (1, 2, 3)] + [b"@x\nACGT\n+\n~~~~\n"]
    for p in parts:
        t = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
        d.append(t, len(p))
        del t  # (the source is free as soon as append returns)
    assert d.pending() == sum(map(len, parts))
    comp = d.flush().tobytes()
    assert d.pending() == 0 and len(d.flush()) == 0
    assert gzip.decompress(comp + ga.BGZF_EOF) == b"".join(parts)
    d.close()
