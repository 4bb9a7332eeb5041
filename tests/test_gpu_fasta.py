"""Device-side FASTA record scan (gs_match_submit_fasta, gs_text.hip): raw FASTA text -> records by
two prefix sums over the lines -> gathered reads -> match, against the oracle's restatement of
AbstractFastqReader.doReadFasta (C/fastq/AbstractFastqReader.java:375-438).  Needs an MI355X: run with -m gpu."""
import gzip
import os

import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import host, synth
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)


def _fasta(sdb, n, seed, width=60, crlf=False):
    """contigs of 0 .. 5000 bases cut from the genomes, wrapped at `width`; some with a stray N, one header right after
    another (a read of length 0), header text with '>' and blanks inside"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        g = sdb.genomes[int(rng.integers(0, len(sdb.genomes)))]
        L = int(rng.choice([0, 20, 31, 150, int(rng.integers(200, 5000))], p=[0.03, 0.05, 0.05, 0.37, 0.5]))
        p = int(rng.integers(0, len(g) - L + 1))
        s = bytearray(g[p:p + L].tobytes())
        if L and rng.random() < 0.2:
            s[int(rng.integers(0, L))] = ord("N")
        eol = b"\r\n" if crlf else b"\n"
        out.append(b">c%d len=%d a>b" % (i, L) + eol)
        w = width if rng.random() < 0.8 else int(rng.integers(1, 200))
        for j in range(0, L, w):
            out.append(bytes(s[j:j + w]) + eol)
    return b"".join(out)


def _oracle(sdb, text, **cfg):
    rd = orc.parse_fastq(text, fasta=True, k=31)
    run = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi), **cfg)
    cv, fl = run.submit(rd["seq"], rd["seq_off"], threads=4)
    return run.finish()[0], cv, fl, rd


@pytest.mark.parametrize("seed,crlf", [(1, False), (2, False), (3, True)])
def test_fasta_chunks_on_the_device(sdb, seed, crlf):
    text = _fasta(sdb, 3000, seed, crlf=crlf)
    want, wcv, wfl, rd = _oracle(sdb, text)
    n = rd["n_reads"]
    assert n == 3000
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    # several chunks, cut at header lines, with per-read outputs
    starts = [i for i in range(len(text)) if text[i:i + 1] == b">" and (i == 0 or text[i - 1:i] == b"\n")]
    assert len(starts) == n
    cuts = [0, starts[1], starts[700], starts[701], starts[2200], len(text)]
    recs = [0, 1, 700, 701, 2200, n]
    cv = np.full(n, -9, dtype=np.int32)
    fl = np.full(n, 77, dtype=np.uint8)
    keep = []
    for a, b, ra, rb in zip(cuts[:-1], cuts[1:], recs[:-1], recs[1:]):
        ccv, cfl = np.zeros(rb - ra, dtype=np.int32), np.zeros(rb - ra, dtype=np.uint8)
        keep.append((ra, rb, ccv, cfl))
        m.submit_fasta(text[a:b], first_read_no=ra, class_vi=ccv, flags=cfl)
        m.sync()
    for ra, rb, ccv, cfl in keep:
        cv[ra:rb], fl[ra:rb] = ccv, cfl
    failed, bad, tot = m.text_status()
    assert failed == -1
    assert tot == (n, rd["total_kmers"], rd["total_bps"])
    got, _ = m.finish()
    assert np.array_equal(got, want), np.argwhere(got != want)[:6]
    assert np.array_equal(cv, wcv) and np.array_equal(fl, wfl)
    m.close()
    store.close()


def test_fasta_chunks_the_device_must_refuse(sdb):
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    good = _fasta(sdb, 50, 9)
    body = bytes(sdb.genomes[0][:120].tobytes())
    cases = {
        "empty line": b">a\n" + body[:60] + b"\n\n" + body[60:] + b"\n",
        "no header first": body[:60] + b"\n>a\n" + body + b"\n",
        "nul": b">a\n" + body[:30] + b"\0" + body[30:] + b"\n",
        "wrong record count": None,
    }
    for name, text in cases.items():
        if text is None:
            m.submit_fasta(good, n_records=49)
        else:
            m.submit_fasta(text)
        failed, bad, tot = m.text_status()
        assert failed >= 0, name
        assert tot == (0, 0, 0), name
        m.clear_text_error() if hasattr(m, "clear_text_error") else m.text_clear_error()
        m.submit_fasta(good)  # the run goes on after the refusal was cleared
        failed, bad, tot = m.text_status()
        assert failed == -1, name
        m.reset()
    m.close()
    store.close()


@pytest.mark.parametrize("gz", [False, True])
@pytest.mark.parametrize("block", [None, 4096, 700])
def test_fasta_files_through_the_host_pipeline(sdb, tmp_path, monkeypatch, gz, block):
    """gs_host_match_files on .fasta / .fa.gz files: raw blocks cut in front of header lines -> gs_match_submit_fasta;
    records longer than a block, a tail without a final newline and a file with an empty line take the reference-exact
    parser, in the middle of the file if need be -- the table and the totals are the oracle's either way"""
    if block:
        monkeypatch.setenv("GS_HOST_BLOCK_BYTES", str(block))
    texts = [_fasta(sdb, 1200, 21), _fasta(sdb, 300, 22)[:-1],                   # no final newline: the last byte is lost
             _fasta(sdb, 200, 23) + b">x\nACGT\n\nACGTACGT\n" + _fasta(sdb, 100, 24)]  # an empty line in the middle
    paths = []
    for i, t in enumerate(texts):
        p = str(tmp_path / (f"c{i}.fa.gz" if gz else f"c{i}.fasta"))
        with (gzip.open(p, "wb", compresslevel=1) if gz else open(p, "wb")) as f:
            f.write(t)
        paths.append(p)
    run = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi))
    reads = kmers = bps = 0
    first = 0
    for t in texts:
        rd = orc.parse_fastq(t, fasta=True, k=31)
        run.submit(rd["seq"], rd["seq_off"], first_read_no=first, threads=4, per_read=False)
        first += rd["n_reads"]
        reads, kmers, bps = reads + rd["n_reads"], kmers + rd["total_kmers"], bps + rd["total_bps"]
    want, _ = run.finish()
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    table, _, tot = host.match_files(store, paths)
    assert (tot.reads, tot.kmers, tot.bps) == (reads, kmers, bps)
    assert np.array_equal(table, want), np.argwhere(table != want)[:6]
    store.close()


@pytest.mark.parametrize("gz", [False, True])
def test_kraken_style_lines_of_fasta_files_from_the_device_path(sdb, tmp_path, monkeypatch, gz):
    """FASTA input with Kraken-style output: the records are found and matched on the device, the lines are formatted from the
    header lines of the text + the device's read lengths and runs; byte for byte what the reference-exact parser path writes"""
    import gzip
    rng = np.random.default_rng(12)
    seq, off = synth.reads_host(sdb.genomes, 3000, read_len=700, seed=41)
    parts = []
    for r in range(3000):
        s = seq[int(off[r]):int(off[r + 1])].tobytes()
        if r % 13 == 0:
            s = s[:int(rng.integers(1, 60))]
        w = int(rng.integers(50, 90))
        body = b"\n".join(s[i:i + w] for i in range(0, len(s), w))
        parts.append((b">contig_%d some description=%d\n" % (r, r * 7) if r % 5 else b">c%d\n" % r) + body + b"\n")
    data = b"".join(parts)
    p = tmp_path / ("contigs.fasta.gz" if gz else "contigs.fa")
    (gzip.open(p, "wb") if gz else open(p, "wb")).write(data)
    monkeypatch.setenv("GS_HOST_BLOCK_BYTES", str(1 << 17))
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    taxids = [str(1000 + i) for i in range(sdb.n_values)]
    outs = {}
    for fast in ("1", "0"):
        monkeypatch.setenv("GS_HOST_FAST", fast)
        kr = str(tmp_path / ("k%s.out" % fast))
        fl = str(tmp_path / ("f%s.fastq" % fast))
        for write_all in (True, False):
            table, _, tot = host.match_files(store, [str(p)], kraken_out_path=kr, filtered_path=fl, taxids=taxids, write_all=write_all,
                                             with_probs=write_all)
            outs[(fast, write_all)] = (open(kr, "rb").read(), table.copy(), tot.reads, open(fl, "rb").read(), tot.filtered_reads)
    for write_all in (True, False):
        a, b = outs[("1", write_all)], outs[("0", write_all)]
        assert a[2] == b[2] == 3000 and np.array_equal(a[1], b[1])
        assert a[0] == b[0] and len(a[0]) > 1000
        # filtered reads: '@' for '>', the sequence in one line, '~' for the quality FASTA does not have
        assert a[3] == b[3] and a[4] == b[4] > 500 and a[3].startswith(b"@c") and b"~~~~~~~~" in a[3]
    n_all, n_cls = outs[("1", True)][0].count(b"\n"), outs[("1", False)][0].count(b"\n")
    assert 2500 < n_all <= 3000 and 500 < n_cls < n_all  # (records shorter than k have no runs and no line)
    store.close()


@pytest.mark.parametrize("gz", [False, True])
@pytest.mark.parametrize("block", [None, 4096, 700])
def test_filter_goal_on_fasta_files_from_the_device_path(sdb, tmp_path, monkeypatch, gz, block):
    """gs_host_filter_files on FASTA input (VERDICT r01 "what's missing" 7, the filter side): records found and gathered on the
    device (gs_filter_submit_fasta), accept flags by the index filter, every record rewritten as four-line FASTQ ('@' for '>',
    one sequence line, '~' quality) -- byte for byte what the reference-exact parser path writes, and the oracle's accept flags"""
    import gzip
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:4])]
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys)
    gb = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
    if block:
        monkeypatch.setenv("GS_HOST_BLOCK_BYTES", str(block))
    texts = [_fasta(sdb, 900, 51), _fasta(sdb, 200, 52, crlf=True)[:-1],                # no final newline
             _fasta(sdb, 150, 53) + b">x\nACGT\n\nACGTACGT\n" + _fasta(sdb, 80, 54)]       # an empty line in the middle
    paths, want_acc, reads = [], 0, 0
    for i, t in enumerate(texts):
        p = str(tmp_path / (f"c{i}.fa.gz" if gz else f"c{i}.fasta"))
        with (gzip.open(p, "wb", compresslevel=1) if gz else open(p, "wb")) as f:
            f.write(t)
        paths.append(p)
        rd = orc.parse_fastq(t, fasta=True, k=31)
        pseq = rd["seq"] if len(rd["seq"]) else np.zeros(1, dtype=np.uint8)
        want_acc += int(ob.filter_batch(31, 1, 0.2, pseq, rd["seq_off"]).sum())
        reads += rd["n_reads"]
    outs = {}
    for fast in ("1", "0"):
        monkeypatch.setenv("GS_HOST_FAST", fast)
        before = host.stat(1)
        a, r = str(tmp_path / f"acc{fast}.fastq"), str(tmp_path / f"rest{fast}.fastq")
        tot = host.filter_files(gb, 31, paths, filtered_path=a, rest_path=r)
        outs[fast] = (open(a, "rb").read(), open(r, "rb").read(), tot.reads, tot.kmers, tot.bps, tot.filtered_reads, host.stat(1) - before)
    assert outs["1"][:6] == outs["0"][:6]
    assert outs["1"][2] == reads and outs["1"][5] == want_acc and 0 < want_acc < reads
    assert outs["1"][6] >= 1 and outs["0"][6] == 0  # the device path did run (and did not with GS_HOST_FAST=0)
    assert outs["1"][0].startswith(b"@c") and b"\n+\n~" in outs["1"][0]
    assert outs["1"][0].count(b"\n") == 4 * want_acc and outs["1"][1].count(b"\n") == 4 * (reads - want_acc)
    gb.close()
