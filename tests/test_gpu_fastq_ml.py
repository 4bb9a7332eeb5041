"""General FASTQ on the device (gs_match_submit_fastq_ml; VERDICT r01 "what's missing" 7): records whose sequence and quality run
over any number of lines (AbstractFastqReader.doReadFastq, :288-368) -- where a record starts depends on everything before it
('@' and '+' may open quality lines), which the device resolves with one "where would the next record start" step per line and
pointer doubling.  Against the parser restatement (orc.parse_fastq, pinned by the reference's SimpleTest.fastq fixture) and the
match tables it leads to.  Needs an MI355X: run with -m gpu."""
import os

import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import host, synth
from conftest import GOLDEN
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(k=31, genera=2, species_per_genus=3, genome_len=20000, seed=33)


def _multiline_fastq(sdb, n, seed, max_seq_lines=4, nasty=True):
    """records as a sequencer would never write them, and the reference still reads them"""
    rng = np.random.default_rng(seed)
    seq, off = synth.reads_host(sdb.genomes, n, read_len=150, seed=seed)
    out = []
    for r in range(n):
        s = seq[int(off[r]):int(off[r + 1])].tobytes()
        if nasty and r % 17 == 3:
            s = b"+" + s[1:]  # the first sequence line may start with '+': it is still sequence
        if nasty and r % 29 == 5:
            s = s[:40]
        desc = b"@r%d some text" % r if (not nasty or r % 11) else b"no at sign %d" % r
        cuts = sorted(set([0, len(s)] + [int(x) for x in rng.integers(1, max(2, len(s)), int(rng.integers(0, max_seq_lines)))]))
        lines = [s[a:b] for a, b in zip(cuts[:-1], cuts[1:])] or [b""]
        lines = [lines[0]] + [l for l in lines[1:] if not l.startswith(b"+")] if len(lines) > 1 else lines
        s2 = b"".join(lines)
        q = bytes(rng.choice(np.frombuffer(b"@+IIIFFF#5~", dtype=np.uint8), len(s2) + (int(rng.integers(0, 5)) if nasty else 0)))
        qc = sorted(set([0, len(q)] + [int(x) for x in rng.integers(1, max(2, len(q)), int(rng.integers(0, 3)))]))
        qlines = [q[a:b] for a, b in zip(qc[:-1], qc[1:])] or [b""]
        # quality is read until it holds at least len(sequence) characters: a split that reaches that early ends the record early
        acc, keep = 0, []
        for ql in qlines:
            keep.append(ql)
            acc += len(ql)
            if acc >= len(s2):
                break
        out.append(desc + b"\n" + b"\n".join(lines) + b"\n+" + (b"anything" if r % 5 == 0 else b"") + b"\n" + b"\n".join(keep) + b"\n")
    return b"".join(out)


def _oracle_table(sdb, data):
    rd = orc.parse_fastq(data, fasta=False, k=31)
    run = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi))
    run.submit(rd["seq"], rd["seq_off"], threads=8, per_read=False)
    return run.finish()[0], rd


@pytest.mark.parametrize("piece", [1 << 30, 50_000, 7_001, 997])
def test_multiline_chunks_against_the_parser_restatement(sdb, piece):
    data = _multiline_fastq(sdb, 3000, seed=5)
    want, rd = _oracle_table(sdb, data)
    assert rd["n_reads"] == 3000  # (the generator and the restatement agree on what a record is)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    carry, done, pos = b"", 0, 0
    while pos < len(data) or carry:
        nxt = data[pos:pos + piece]
        pos += len(nxt)
        buf = carry + nxt
        cut = buf.rfind(b"\n") + 1
        chunk, rest = buf[:cut], buf[cut:]
        if chunk:
            n_rec, used = m.submit_fastq_ml(np.frombuffer(chunk, dtype=np.uint8), first_read_no=done)
            done += n_rec
            carry = chunk[used:] + rest
        else:
            carry = buf
        if pos >= len(data):
            # what is left is an incomplete record (or nothing): the general parser's share, as in the file pipeline
            if carry:
                tail = orc.parse_fastq(carry, fasta=False, k=31)
                if tail["n_reads"]:
                    m.submit(tail["seq"], tail["seq_off"], done, n_reads=tail["n_reads"])
                    done += tail["n_reads"]
            break
    m.sync()
    failed, bad, totals = m.text_status()
    assert failed < 0
    table = m.finish()[0]
    assert done == 3000
    assert np.array_equal(table, want), np.argwhere(table != want)[:6]
    m.close()
    store.close()


def test_reference_fixture_and_refusals(sdb):
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    raw = open(os.path.join(GOLDEN, "fastq", "SimpleTest.fastq"), "rb").read()
    rd = orc.parse_fastq(raw, fasta=False, k=31)
    cut = raw.rfind(b"\n") + 1
    n_rec, used = m.submit_fastq_ml(np.frombuffer(raw[:cut], dtype=np.uint8))
    m.sync()
    failed, bad, totals = m.text_status()
    assert failed < 0
    # every record of the fixture that ends inside the text is found, with the lengths the parser restatement gives
    assert n_rec in (rd["n_reads"], rd["n_reads"] - 1) and used <= cut
    lens = (rd["seq_off"][1:] - rd["seq_off"][:-1])[:n_rec]
    assert totals[0] == n_rec and totals[2] == int(lens.sum())
    # a NUL byte: the reference drops it from the line, the device refuses the chunk (the host parser takes the file)
    bad_text = b"@a\nACGT\x00ACGT\n+\nIIIIIIII\n"
    m.reset()
    n_rec, used = m.submit_fastq_ml(np.frombuffer(bad_text, dtype=np.uint8))
    m.sync()
    failed, _, _ = m.text_status()
    assert n_rec == 0 and used == 0 and failed >= 0
    m.close()
    store.close()


@pytest.mark.parametrize("gz", [False, True])
def test_multiline_fastq_files_go_through_the_device(sdb, tmp_path, monkeypatch, gz):
    """gs_host_match_files on a FASTQ file that is not four lines per record: the four-line scan refuses the first chunk, the file
    is read again with the records found on the device, and what that leaves at the end goes to the reference-exact parser"""
    import gzip
    monkeypatch.setenv("GS_HOST_BLOCK_BYTES", str(1 << 16))  # many chunks, records across every block border
    data = _multiline_fastq(sdb, 6000, seed=9) + b"@last without a newline at the end\nACGTACGTAC\nGT\n+\nIIIIIIIIIIII"
    want, rd = _oracle_table(sdb, data)
    p = tmp_path / ("ml.fastq.gz" if gz else "ml.fastq")
    (gzip.open(p, "wb") if gz else open(p, "wb")).write(data)
    plain = tmp_path / "plain.fastq"  # a four-line file in the same call: the usual path, untouched
    rs, ro = synth.reads_host(sdb.genomes, 500, read_len=150, seed=77)
    with open(plain, "wb") as f:
        for i in range(500):
            s = rs[int(ro[i]):int(ro[i + 1])].tobytes()
            f.write(b"@p%d\n%s\n+\n%s\n" % (i, s, b"F" * len(s)))
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    before = host.stat(0)
    table, _, tot = host.match_files(store, [str(p)])
    assert host.stat(0) > before + 5  # chunks DID go through the device's record search
    assert np.array_equal(table, want), np.argwhere(table != want)[:6]
    assert (tot.reads, tot.bps, tot.kmers) == (rd["n_reads"], int(rd["total_bps"]), int(rd["total_kmers"]))
    # the same through the reference-exact parser alone, and mixed with a four-line file
    monkeypatch.setenv("GS_HOST_ML", "0")
    t2, _, tot2 = host.match_files(store, [str(p)])
    monkeypatch.delenv("GS_HOST_ML")
    assert np.array_equal(t2, want) and tot2.reads == tot.reads
    both = _oracle_table(sdb, data + b"\n" + open(plain, "rb").read())[0]
    t3, _, tot3 = host.match_files(store, [str(p), str(plain)])
    assert tot3.reads == rd["n_reads"] + 500
    assert np.array_equal(t3[:, [0, 1, 2, 3, 4, 5]], both[:, [0, 1, 2, 3, 4, 5]])
    store.close()


def test_kraken_style_lines_of_multiline_fastq_from_the_device_path(sdb, tmp_path, monkeypatch):
    """multi-line FASTQ with Kraken-style output: records found and matched on the device, lines from the descriptor lines the
    device classified + its read lengths and runs; byte for byte the parser path's output"""
    monkeypatch.setenv("GS_HOST_BLOCK_BYTES", str(1 << 16))
    data = _multiline_fastq(sdb, 5000, seed=23) + b"@tail\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n"
    p = tmp_path / "ml.fastq"
    p.write_bytes(data)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    taxids = [str(500 + i) for i in range(sdb.n_values)]
    outs = {}
    for ml in ("1", "0"):
        monkeypatch.setenv("GS_HOST_ML", ml)
        before = host.stat(0)
        kr = str(tmp_path / ("k%s.out" % ml))
        table, _, tot = host.match_files(store, [str(p)], kraken_out_path=kr, taxids=taxids, write_all=True)
        outs[ml] = [open(kr, "rb").read(), table.copy(), tot.reads, host.stat(0) - before]
        for probs in (False, True):  # filtered reads, with '~' and with the record's own quality lines
            fl = str(tmp_path / ("f%s_%d.fastq" % (ml, probs)))
            _, _, tot = host.match_files(store, [str(p)], filtered_path=fl, with_probs=probs)
            outs[ml] += [open(fl, "rb").read(), tot.filtered_reads]
    assert outs["1"][3] > 3 and outs["0"][3] == 0  # the device path did run (and did not with GS_HOST_ML=0)
    assert outs["1"][2] == outs["0"][2] == 5001 and np.array_equal(outs["1"][1], outs["0"][1])
    assert outs["1"][0] == outs["0"][0] and outs["1"][0].count(b"\n") > 4000
    assert outs["1"][4] == outs["0"][4] and outs["1"][5] == outs["0"][5] > 1000 and b"~~~~" in outs["1"][4]
    assert outs["1"][6] == outs["0"][6] and outs["1"][7] == outs["0"][7] and outs["1"][6] != outs["1"][4]
    store.close()


@pytest.mark.parametrize("gz", [False, True])
def test_filter_goal_on_multiline_fastq_from_the_device_path(sdb, tmp_path, monkeypatch, gz):
    """gs_host_filter_files on a FASTQ file that is not four lines per record: the four-line scan refuses the first chunk, the file
    is read again with the records found on the device (gs_filter_submit_fastq_ml); accepted / rejected reads are rewritten as
    four-line FASTQ, with '~' or with the record's own quality lines -- byte for byte the parser path's files, and the oracle's
    accept flags"""
    import gzip
    monkeypatch.setenv("GS_HOST_BLOCK_BYTES", str(1 << 16))
    data = _multiline_fastq(sdb, 5000, seed=19) + b"@last without a newline at the end\nACGTACGTAC\nGT\n+\nIIIIIIIIIIII"
    p = tmp_path / ("ml.fastq.gz" if gz else "ml.fastq")
    (gzip.open(p, "wb") if gz else open(p, "wb")).write(data)
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:3])]
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys)
    gb = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
    rd = orc.parse_fastq(data, fasta=False, k=31)
    want = ob.filter_batch(31, 1, 0.2, rd["seq"], rd["seq_off"])
    outs = {}
    for ml in ("1", "0"):
        monkeypatch.setenv("GS_HOST_ML", ml)
        for probs in (False, True):
            before = host.stat(1)
            a, r = str(tmp_path / f"acc{ml}{probs}.fastq"), str(tmp_path / f"rest{ml}{probs}.fastq")
            tot = host.filter_files(gb, 31, [str(p)], filtered_path=a, rest_path=r, with_probs=probs)
            outs[(ml, probs)] = (open(a, "rb").read(), open(r, "rb").read(), tot.reads, tot.kmers, tot.bps, tot.filtered_reads,
                                 host.stat(1) - before)
    for probs in (False, True):
        a, b = outs[("1", probs)], outs[("0", probs)]
        assert a[:6] == b[:6]
        assert a[6] > 3 and b[6] == 0  # the device path did run (and did not with GS_HOST_ML=0)
        assert (a[2], a[3], a[4]) == (rd["n_reads"], int(rd["total_kmers"]), int(rd["total_bps"]))
        assert a[5] == int(want.sum()) and 500 < a[5] < rd["n_reads"]
        assert a[0].count(b"\n") == 4 * a[5] and a[1].count(b"\n") == 4 * (rd["n_reads"] - a[5])
    assert outs[("1", True)][0] != outs[("1", False)][0] and b"\n+\n~~~~" in outs[("1", False)][0]
    gb.close()


def test_filter_chunks_of_general_fastq_and_fasta_through_the_abi(sdb):
    """gs_filter_submit_fastq_ml / gs_filter_submit_fasta chunk by chunk: accept flags and read lengths of every record against the
    parser restatement + the oracle's filter; a chunk with a NUL byte is refused (n_records = -1) and changes nothing"""
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:3])]
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys)
    gb = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
    f = ga.FastqBloomFilter(31, gb)
    data = _multiline_fastq(sdb, 2500, seed=41)
    rd = orc.parse_fastq(data, fasta=False, k=31)
    want = ob.filter_batch(31, 1, 0.2, rd["seq"], rd["seq_off"])
    got, lens, carry, pos, piece = [], [], b"", 0, 30011
    while pos < len(data):
        buf = carry + data[pos:pos + piece]
        pos += piece
        cut = buf.rfind(b"\n") + 1
        acc = np.zeros(buf.count(b"\n") // 4 + 2, dtype=np.uint8)
        n_rec, used, ln = f.submit_fastq_ml(buf[:cut], acc)
        assert n_rec >= 0
        got.append(acc[:n_rec].copy())
        lens.append(ln)
        carry = buf[used:]
    assert carry == b""  # (the text ends with a newline behind a complete record)
    got, lens = np.concatenate(got), np.concatenate(lens)
    assert len(got) == 2500 and np.array_equal(got, want) and 100 < int(want.sum()) < 2500
    assert np.array_equal(lens, np.diff(rd["seq_off"].astype(np.int64)))
    assert f.text_status()[2] == (2500, int(rd["total_kmers"]), int(rd["total_bps"]))
    acc = np.full(8, 7, dtype=np.uint8)
    n_rec, used, _ = f.submit_fastq_ml(b"@a\nACGT\0ACGT\n+\nIIIIIIIII\n", acc)
    assert n_rec == -1 and f.text_status()[0] >= 0
    f.text_reset(clear_totals=True)
    # FASTA: wrapped records, one of length 0
    g = sdb.genomes[0].tobytes()
    recs = [g[i * 700:i * 700 + L] for i, L in enumerate([150, 0, 31, 30, 2000, 64, 500])]
    text = b"".join(b">r%d x\n" % i + b"".join(s[j:j + 70] + b"\n" for j in range(0, len(s), 70)) for i, s in enumerate(recs))
    acc = np.zeros(len(recs), dtype=np.uint8)
    ln = f.submit_fasta(text, acc)
    rd = orc.parse_fastq(text, fasta=True, k=31)
    assert list(ln) == [len(s) for s in recs] == list(np.diff(rd["seq_off"].astype(np.int64)))
    assert np.array_equal(acc, ob.filter_batch(31, 1, 0.2, rd["seq"], rd["seq_off"]))
    assert f.text_status()[0] < 0
    gb.close()
