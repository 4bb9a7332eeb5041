"""End-to-end file pipelines of the C++ host layer on the GPU (gs_host_match_files / gs_host_filter_files):
FASTQ(.gz) files -> per-taxid table, totals, Kraken-style output, filtered FASTQ -- against the oracle."""
import gzip
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
    return synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)


def _write_fastq(path, seq, off, gz, start=0):
    n = len(off) - 1
    recs = []
    for i in range(n):
        s = seq[int(off[i]):int(off[i + 1])].tobytes()
        recs.append(b"@r%d sample=%d\n%s\n+\n%s\n" % (start + i, i % 7, s, b"F" * len(s)))
    data = b"".join(recs)
    with (gzip.open(path, "wb", compresslevel=1) if gz else open(path, "wb")) as f:
        f.write(data)


def _oracle_kraken_lines(odb, taxids, seq, off, cv, start=0, write_all=True):
    lines = []
    names = {-1: "0", -2: "A"}
    for i in range(len(off) - 1):
        r = seq[int(off[i]):int(off[i + 1])].tobytes()
        segs = odb.segments(r)
        if not segs or not (write_all or cv[i] >= 0):
            continue
        segtxt = " ".join(f"{taxids[c] if c >= 0 else names[c]}:{n}" for c, n in segs)
        lines.append(f"{'C' if cv[i] >= 0 else 'U'}\tr{start + i}\t{taxids[cv[i]] if cv[i] >= 0 else '0'}\t{len(r)}\t{segtxt}")
    return lines


def test_match_files_two_inputs_table_kraken_filtered(sdb, tmp_path):
    seq, off = synth.reads_host(sdb.genomes, 7000, read_len=150, seed=19)
    seq = seq.copy()
    seq[150 * 5 + 70] = ord("N")
    seq[150 * 6:150 * 6 + 3] = ord("N")
    f1, f2 = str(tmp_path / "a.fastq.gz"), str(tmp_path / "b.fq")
    _write_fastq(f1, seq, off[:4001], gz=True)
    off2 = off[4000:] - off[4000]
    _write_fastq(f2, seq[int(off[4000]):], off2, gz=False, start=4000)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    kr, flt = str(tmp_path / "kraken.out"), str(tmp_path / "filtered.fastq.gz")
    table, dtable, tot = host.match_files(store, [f1, f2], filtered_path=flt, kraken_out_path=kr, taxids=sdb.taxids,
                                          batch_reads=1500)
    # oracle: same reads, global read numbers running over both files
    odb = orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    orun = orc.MatchRun(odb)
    ocv, ofl = orun.submit(seq, off)
    otable, _ = orun.finish()
    assert np.array_equal(table, otable)
    assert (tot.reads, tot.kmers, tot.bps) == (7000, 7000 * 120, 7000 * 150)
    want = _oracle_kraken_lines(odb, sdb.taxids, seq, off, ocv)
    got = open(kr).read().rstrip("\n").split("\n")
    assert got == want
    # filtered FASTQ: reads with matchRead() == true, rewritten with '~' qualities (withProbs = false)
    flt_txt = gzip.open(flt).read().decode().split("\n")
    exp = []
    for i in np.flatnonzero(ofl & orc.F_RETURNED):
        s = seq[int(off[i]):int(off[i + 1])].tobytes().decode()
        exp += [f"@r{i} sample={(i if i < 4000 else i - 4000) % 7}", s, "+", "~" * len(s)]
    assert flt_txt[:-1] == exp and tot.filtered_reads == int((ofl & orc.F_RETURNED != 0).sum())
    store.close()


def test_match_files_write_all_false_and_dengue_golden(tmp_path):
    lines = open(os.path.join(GOLDEN, "dengue1", "dengue1.fasta")).read().split("\n")
    genome = "".join(l.strip() for l in lines if not l.startswith(">")).upper()
    keys = np.unique(orc.canonical_kmers(genome, 31))
    store = ga.DeviceKMerStore(31, keys, np.zeros(len(keys), np.int32), 1, np.array([-1], np.int32))
    junk = tmp_path / "mixed.fastq"
    junk.write_bytes(open(os.path.join(GOLDEN, "dengue1", "test.fastq"), "rb").read() +
                     b"@nohit\n" + b"ACGT" * 12 + b"\n+\n" + b"I" * 48 + b"\n@tiny\nACG\n+\nIII\n")
    kr = str(tmp_path / "k.out")
    table, _, tot = host.match_files(store, [str(junk)], kraken_out_path=kr, taxids=["1"], write_all=False)
    assert open(kr).read() == open(os.path.join(GOLDEN, "dengue1", "test.out")).read()
    kr2 = str(tmp_path / "k2.out")
    host.match_files(store, [str(junk)], kraken_out_path=kr2, taxids=["1"], write_all=True)
    assert open(kr2).read().split("\n")[:2] == ["C\ttest\t1\t41\t0:2 1:7 0:2", "U\tnohit\t0\t48\t0:18"]
    assert tot.reads == 3 and table[0, 0] == 1
    # the reference's own FASTA fixture goes through the FASTA path (suffix) and every k-mer is found
    t2, _, tot2 = host.match_files(store, [os.path.join(GOLDEN, "dengue1", "dengue1.fasta")])
    assert tot2.reads == 1 and tot2.bps == 10735 and t2[0, 2] == 10705 and t2[0, 3] == len(keys)
    store.close()


def test_filter_files(sdb, tmp_path):
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:4])]
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys)
    gb = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
    seq, off = synth.reads_host(sdb.genomes, 5000, read_len=150, seed=5)
    f1 = str(tmp_path / "in.fastq.gz")
    _write_fastq(f1, seq, off, gz=True)
    acc_p, rest_p = str(tmp_path / "acc.fastq"), str(tmp_path / "rest.fastq.gz")
    tot = host.filter_files(gb, 31, [f1], filtered_path=acc_p, rest_path=rest_p)
    want = ob.filter_batch(31, 1, 0.2, seq, off)
    assert tot.reads == 5000 and tot.filtered_reads == int(want.sum())
    acc_ids = [l for l in open(acc_p).read().split("\n") if l.startswith("@r")]
    rest_ids = [l for l in gzip.open(rest_p).read().decode().split("\n") if l.startswith("@r")]
    assert acc_ids == [f"@r{i} sample={i % 7}" for i in np.flatnonzero(want)]
    assert rest_ids == [f"@r{i} sample={i % 7}" for i in np.flatnonzero(want == 0)]
    gb.close()


# ------------------------------------------------------------------ text fast path (device-side record scan)
def _oracle_file(sdb, data):
    p = orc.parse_fastq(data, k=31)
    seq = p["seq"] if len(p["seq"]) else np.zeros(1, dtype=np.uint8)
    odb = orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    run = orc.MatchRun(odb)
    run.submit(seq, p["seq_off"])
    t, _ = run.finish()
    return t, (int(p["n_reads"]), int(p["total_kmers"]), int(p["total_bps"]))


def _fastq_bytes(sdb, n, seed, nl=b"\n"):
    seq, off = synth.reads_host(sdb.genomes, n, read_len=150, seed=seed)
    recs = []
    for i in range(n):
        s = seq[int(off[i]):int(off[i + 1])].tobytes()
        if i % 11 == 3:
            s = s[:i % 40]                    # short / empty reads
        if i % 13 == 5:
            s = s[:60] + b"N" + s[61:]
        if i % 17 == 7:
            s = s * 3                         # long reads
        recs.append(b"@r%d x\n".replace(b"\n", nl) % i + s + nl + b"+" + nl + b"F" * len(s) + nl)
    return recs


@pytest.mark.parametrize("block", [0, 64, 1000, 4096])
@pytest.mark.parametrize("shape", ["plain", "crlf", "no final newline", "multi-line record", "truncated", "nul", "tiny"])
def test_match_files_text_path_equals_reference_parser(sdb, tmp_path, monkeypatch, block, shape):
    """every file shape must give the general parser's result; the device path may only be faster (block = 0: default
    32 MiB blocks, else tiny blocks so that records straddle block boundaries and carries pile up)"""
    recs = _fastq_bytes(sdb, 600, seed=23, nl=b"\r\n" if shape == "crlf" else b"\n")
    if shape == "multi-line record":
        s = recs[400].split(b"\n")
        recs[400] = s[0] + b"\n" + s[1][:50] + b"\n" + s[1][50:] + b"\n+\n" + s[3] + b"\n"
    data = b"".join(recs)
    if shape == "no final newline":
        data = data[:-1]
    elif shape == "truncated":
        data = data[:-200]
    elif shape == "nul":
        data = data[:30000] + b"\0" + data[30000:]
    elif shape == "tiny":
        data = b"".join(recs[:1])
    path = str(tmp_path / "in.fastq")
    open(path, "wb").write(data)
    want_t, want_tot = _oracle_file(sdb, data)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    if block:
        monkeypatch.setenv("GS_HOST_BLOCK_BYTES", str(block))
    table, _, tot = host.match_files(store, [path])
    assert (tot.reads, tot.kmers, tot.bps) == want_tot
    assert np.array_equal(table, want_t), np.argwhere(table != want_t)[:8]
    monkeypatch.setenv("GS_HOST_FAST", "0")
    table2, _, tot2 = host.match_files(store, [path])
    assert np.array_equal(table2, want_t) and (tot2.reads, tot2.kmers, tot2.bps) == want_tot
    store.close()


def test_match_files_text_path_several_files_and_max_contig_order(sdb, tmp_path):
    """read numbers run over all files of a call, also when text chunks and parsed batches alternate"""
    recs = _fastq_bytes(sdb, 900, seed=29)
    parts = [b"".join(recs[:300]), b"".join(recs[300:600]), b"".join(recs[600:])]
    paths = [str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq.gz"), str(tmp_path / "c.fq")]
    open(paths[0], "wb").write(parts[0])
    with gzip.open(paths[1], "wb") as f:
        f.write(parts[1])
    open(paths[2], "wb").write(parts[2])
    want_t, want_tot = _oracle_file(sdb, b"".join(parts))
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    table, _, tot = host.match_files(store, paths)
    assert np.array_equal(table, want_t)
    assert (tot.reads, tot.kmers, tot.bps) == want_tot
    store.close()


def test_match_files_empty_file(sdb, tmp_path):
    path = str(tmp_path / "empty.fastq")
    open(path, "wb").close()
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    table, _, tot = host.match_files(store, [path])
    assert tot.reads == 0 and not table[:, :9].any()
    store.close()


def test_device_blocks_kept_between_calls_and_trimmed(sdb, tmp_path, monkeypatch):
    """The C ABI library keeps the device buffers of a finished run for the next one (gs_api.cpp's block cache): calls of the same and
    of other shapes one after the other, with the cache trimmed in between (gs_host_release_pools -> gs_device_cache_trim), give the
    tables of the first call -- a block is handed out again only when the device is idle and is never zeroed, so a run that counted on
    fresh memory would show here."""
    seq, off = synth.reads_host(sdb.genomes, 6000, read_len=150, seed=23)
    big, small = str(tmp_path / "big.fastq"), str(tmp_path / "small.fastq.gz")
    _write_fastq(big, seq, off, gz=False)
    _write_fastq(small, seq[:int(off[700])], off[:701], gz=True)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    want_big = host.match_files(store, [big])[0]
    want_small = host.match_files(store, [small])[0]
    assert want_big[:, :9].any() and want_small[:, :9].any() and not np.array_equal(want_big, want_small)
    for rnd in range(3):
        assert np.array_equal(host.match_files(store, [big])[0], want_big)
        assert np.array_equal(host.match_files(store, [small], kraken_out_path=str(tmp_path / "k.out"), taxids=sdb.taxids)[0], want_small)
        assert np.array_equal(host.match_files(store, [big], batch_reads=900)[0], want_big)
        if rnd == 1:
            host.release_pools()
            assert ga.lib().gs_device_cache_trim() == 0
    store.close()
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)  # (its buffers may be the old store's)
    assert np.array_equal(host.match_files(store, [big])[0], want_big)
    store.close()


@pytest.mark.parametrize("block", [0, 100, 5000])
@pytest.mark.parametrize("shape", ["plain", "crlf", "no final newline", "multi-line record", "nul"])
def test_filter_files_text_path_equals_reference_parser(sdb, tmp_path, monkeypatch, block, shape):
    """gs_host_filter_files over plain FASTQ: the device-side record scan must give byte-identical output files
    and totals to the reference-exact parser path, whatever the file shape and the block size"""
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:4])]
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys)
    gb = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
    recs = _fastq_bytes(sdb, 500, seed=31, nl=b"\r\n" if shape == "crlf" else b"\n")
    if shape == "multi-line record":
        s = recs[300].split(b"\n")
        recs[300] = s[0] + b"\n" + s[1][:50] + b"\n" + s[1][50:] + b"\n+\n" + s[3] + b"\n"
    data = b"".join(recs)
    if shape == "no final newline":
        data = data[:-1]
    elif shape == "nul":
        data = data[:20000] + b"\0" + data[20000:]
    path = str(tmp_path / "in.fq")
    open(path, "wb").write(data)
    p = orc.parse_fastq(data, k=31)
    pseq = p["seq"] if len(p["seq"]) else np.zeros(1, dtype=np.uint8)
    want = ob.filter_batch(31, 1, 0.2, pseq, p["seq_off"])
    outs = {}
    for fast in ("1", "0"):
        monkeypatch.setenv("GS_HOST_FAST", fast)
        if block:
            monkeypatch.setenv("GS_HOST_BLOCK_BYTES", str(block))
        a, r = str(tmp_path / f"acc{fast}.fastq"), str(tmp_path / f"rest{fast}.fastq")
        tot = host.filter_files(gb, 31, [path], filtered_path=a, rest_path=r)
        assert (tot.reads, tot.kmers, tot.bps) == (int(p["n_reads"]), int(p["total_kmers"]), int(p["total_bps"]))
        assert tot.filtered_reads == int(want.sum())
        outs[fast] = (open(a, "rb").read(), open(r, "rb").read())
    assert outs["1"] == outs["0"]
    # and the accepted file holds exactly the oracle's accepted reads, rewritten like ReadEntry.write
    exp = b""
    for i in np.flatnonzero(want):
        d = bytes(p["desc"][int(p["desc_off"][i]):int(p["desc_off"][i + 1])])
        s = bytes(p["seq"][int(p["seq_off"][i]):int(p["seq_off"][i + 1])])
        exp += d + b"\n" + s + b"\n+\n" + b"~" * len(s) + b"\n"
    assert outs["1"][0] == exp
    gb.close()


@pytest.mark.parametrize("container", ["bgzf", "gzip"])
@pytest.mark.parametrize("text_target", [0, 70000])
@pytest.mark.parametrize("shape", ["plain", "crlf", "no final newline", "multi-line record", "multi-line first"])
def test_filter_files_bgzf_input_inflated_on_the_device(sdb, tmp_path, monkeypatch, shape, text_target, container):
    """gs_host_filter_files over block-gzip FASTQ: members inflated on the device, the filter on the device text, the text back once
    for the writers (filter_bgzf_file) -- output files and totals byte-identical to the host-decoder path (GS_DEVICE_INFLATE=0)
    and to the oracle, whatever the file shape; small feeds (GS_HOST_BGZF_TEXT) put record and member boundaries everywhere"""
    from conftest import bgzf
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:4])]
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys)
    gb = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
    recs = _fastq_bytes(sdb, 1500, seed=41, nl=b"\r\n" if shape == "crlf" else b"\n")
    for at in ([900] if shape == "multi-line record" else [0] if shape == "multi-line first" else []):
        s = recs[at].split(b"\n")
        recs[at] = s[0] + b"\n" + s[1][:50] + b"\n" + s[1][50:] + b"\n+\n" + s[3] + b"\n"
    data = b"".join(recs)
    if shape == "no final newline":
        data = data[:-1]
    path = str(tmp_path / "in.fastq.gz")
    if container == "bgzf":
        open(path, "wb").write(bgzf(data, block=20000, level=1))
    else:  # one gzip member: inflated on the device as a whole (gs_gunzip_plan_device), handed on in slices
        import gzip
        open(path, "wb").write(gzip.compress(data, compresslevel=6, mtime=0))
        monkeypatch.setenv("GS_GUNZIP_CHUNK", "4096")
        if text_target:
            monkeypatch.setenv("GS_GUNZIP_SLOTS", "5")  # (the stream in many batches)
    p = orc.parse_fastq(data, k=31)
    want = ob.filter_batch(31, 1, 0.2, p["seq"], p["seq_off"])
    if text_target:
        monkeypatch.setenv("GS_HOST_BGZF_TEXT", str(text_target))
    outs = {}
    for dev in ("1", "0"):
        monkeypatch.setenv("GS_DEVICE_INFLATE", dev)
        a, r = str(tmp_path / f"acc{dev}.fastq"), str(tmp_path / f"rest{dev}.fastq")
        tot = host.filter_files(gb, 31, [path], filtered_path=a, rest_path=r)
        assert (tot.reads, tot.kmers, tot.bps) == (int(p["n_reads"]), int(p["total_kmers"]), int(p["total_bps"])), dev
        assert tot.filtered_reads == int(want.sum()), dev
        outs[dev] = (open(a, "rb").read(), open(r, "rb").read())
    assert outs["1"] == outs["0"]
    exp = b""
    for i in np.flatnonzero(want):
        d = bytes(p["desc"][int(p["desc_off"][i]):int(p["desc_off"][i + 1])])
        s = bytes(p["seq"][int(p["seq_off"][i]):int(p["seq_off"][i + 1])])
        exp += d + b"\n" + s + b"\n+\n" + b"~" * len(s) + b"\n"
    assert outs["1"][0] == exp
    gb.close()


@pytest.mark.parametrize("odd", [False, True])
def test_match_files_gzip_files_side_by_side(sdb, tmp_path, monkeypatch, odd):
    """two or more gzip files are inflated side by side (each behind its own thread, own status bank on the device,
    read numbers file << 32 | read); the table -- including the max-contig read numbers -- must equal the sequential run"""
    recs = _fastq_bytes(sdb, 1200, seed=37)
    if odd:  # one file needs the general parser from its middle on, one has no final newline
        s = recs[700].split(b"\n")
        recs[700] = s[0] + b"\n" + s[1][:40] + b"\n" + s[1][40:] + b"\n+\n" + s[3] + b"\n"
    cuts = [0, 250, 251, 600, 900, 1200]
    parts = [b"".join(recs[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    if odd:
        parts[1] = parts[1][:-1]
    paths = []
    for i, part in enumerate(parts):
        p = str(tmp_path / f"part{i}.fastq.gz")
        if i == 2:
            p = str(tmp_path / "part2.fastq")  # a plain file among the gzip files
            open(p, "wb").write(part)
        else:
            with gzip.open(p, "wb", compresslevel=1 + i) as f:
                f.write(part)
        paths.append(p)
    want_t, want_tot = _oracle_file(sdb, b"".join(p + (b"\n" if odd and i == 1 else b"") for i, p in enumerate(parts)))
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    monkeypatch.setenv("GS_HOST_BLOCK_BYTES", str(70000))
    t1, _, tot1 = host.match_files(store, paths)
    monkeypatch.setenv("GS_HOST_PARALLEL_FILES", "0")
    t0, _, tot0 = host.match_files(store, paths)
    assert np.array_equal(t1, t0), np.argwhere(t1 != t0)[:8]
    assert (tot1.reads, tot1.kmers, tot1.bps) == (tot0.reads, tot0.kmers, tot0.bps) == want_tot
    assert np.array_equal(t0, want_t), np.argwhere(t0 != want_t)[:8]
    store.close()


@pytest.mark.parametrize("block", [0, 3000])
@pytest.mark.parametrize("shape", ["plain", "crlf", "no final newline", "multi-line record", "gzip"])
def test_match_files_text_path_per_read_outputs(sdb, tmp_path, monkeypatch, block, shape):
    """filtered FASTQ and Kraken-style lines written from the raw blocks (record geometry and segments come from the
    device) must be byte-identical to what the reference-exact parser path writes"""
    recs = _fastq_bytes(sdb, 700, seed=41, nl=b"\r\n" if shape == "crlf" else b"\n")
    if shape == "multi-line record":
        s = recs[500].split(b"\n")
        recs[500] = s[0] + b"\n" + s[1][:50] + b"\n" + s[1][50:] + b"\n+\n" + s[3] + b"\n"
    data = b"".join(recs)
    if shape == "no final newline":
        data = data[:-1]
    path = str(tmp_path / ("in.fastq.gz" if shape == "gzip" else "in.fastq"))
    with (gzip.open(path, "wb") if shape == "gzip" else open(path, "wb")) as f:
        f.write(data)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    outs = {}
    for fast in ("1", "0"):
        monkeypatch.setenv("GS_HOST_FAST", fast)
        if block:
            monkeypatch.setenv("GS_HOST_BLOCK_BYTES", str(block))
        kr, fl = str(tmp_path / f"k{fast}.out"), str(tmp_path / f"f{fast}.fastq")
        t, _, tot = host.match_files(store, [path], filtered_path=fl, kraken_out_path=kr, taxids=sdb.taxids, write_all=(block == 0))
        outs[fast] = (t, (tot.reads, tot.kmers, tot.bps, tot.filtered_reads), open(kr, "rb").read(), open(fl, "rb").read())
    assert np.array_equal(outs["1"][0], outs["0"][0])
    assert outs["1"][1] == outs["0"][1]
    assert outs["1"][2] == outs["0"][2] and len(outs["1"][2]) > 1000
    assert outs["1"][3] == outs["0"][3] and len(outs["1"][3]) > 1000
    store.close()


def test_match_files_corrupt_gzip_is_an_error(sdb, tmp_path):
    """a damaged or truncated gzip file must fail the call (GZIPInputStream throws), not yield a silently shorter run"""
    data = b"".join(_fastq_bytes(sdb, 3000, seed=43))
    good = str(tmp_path / "good.fastq.gz")
    with gzip.open(good, "wb") as f:
        f.write(data)
    raw = open(good, "rb").read()
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    t, _, tot = host.match_files(store, [good])
    assert tot.reads == 3000
    flipped = bytearray(raw)
    flipped[len(raw) // 2] ^= 0x20
    for name, blob in (("flip", bytes(flipped)), ("cut", raw[:len(raw) // 2]), ("crc", raw[:-8] + bytes([raw[-8] ^ 1]) + raw[-7:])):
        p = str(tmp_path / f"{name}.fastq.gz")
        open(p, "wb").write(blob)
        with pytest.raises(RuntimeError):
            host.match_files(store, [p])
    store.close()


@pytest.mark.parametrize("fast", ["1", "0"])
def test_with_probs_keeps_the_quality_lines(sdb, tmp_path, monkeypatch, fast):
    """withProbs (GSConfigKey WITH_PROBS): ReadEntry.write puts out the quality characters it read -- the whole line
    when it is longer than the read, several lines joined when the record has them -- instead of '~' x length.
    The file starts as plain four-line FASTQ (device text path) and turns multi-line later (reference parser)."""
    monkeypatch.setenv("GS_HOST_FAST", fast)
    monkeypatch.setenv("GS_HOST_BLOCK_BYTES", "65536")
    seq, off = synth.reads_host(sdb.genomes, 3000, read_len=150, seed=59)
    rng = np.random.default_rng(4)
    recs, reads, quals = [], [], []
    for i in range(3000):
        r = seq[int(off[i]):int(off[i + 1])].tobytes()
        q = bytes(rng.integers(33, 74, len(r) + (3 if i % 9 == 0 else 0), dtype=np.uint8))  # some longer than the read
        if i >= 2000 and i % 5 == 0:   # sequence and quality split over two lines each
            recs.append(b"@r%d\n" % i + r[:70] + b"\n" + r[70:] + b"\n+\n" + q[:50] + b"\n" + q[50:] + b"\n")
        else:
            recs.append(b"@r%d\n" % i + r + b"\n+\n" + q + b"\n")
        reads.append(r)
        quals.append(q)
    path = str(tmp_path / "in.fastq")
    open(path, "wb").write(b"".join(recs))
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    odb = orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    orun = orc.MatchRun(odb)
    _, ofl = orun.submit(seq, off)
    orun.finish()
    keep = np.flatnonzero(ofl & orc.F_RETURNED)
    assert 500 < len(keep)
    flt = str(tmp_path / "f.fastq")
    _, _, tot = host.match_files(store, [path], filtered_path=flt, with_probs=True)
    want = b"".join(b"@r%d\n" % i + reads[i] + b"\n+\n" + quals[i] + b"\n" for i in keep)
    assert open(flt, "rb").read() == want and tot.filtered_reads == len(keep)
    _, _, tot = host.match_files(store, [path], filtered_path=flt)  # the default stays '~' x length
    want = b"".join(b"@r%d\n" % i + reads[i] + b"\n+\n" + b"~" * len(reads[i]) + b"\n" for i in keep)
    assert open(flt, "rb").read() == want
    # the filter goal has the same switch
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:4])]
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys)
    bloom = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
    acc, rest = str(tmp_path / "acc.fastq"), str(tmp_path / "rest.fastq")
    host.filter_files(bloom, 31, [path], filtered_path=acc, rest_path=rest, with_probs=True)
    ok = ob.filter_batch(31, 1, 0.2, seq, off)
    for out, sel in ((acc, np.flatnonzero(ok)), (rest, np.flatnonzero(ok == 0))):
        assert open(out, "rb").read() == b"".join(b"@r%d\n" % i + reads[i] + b"\n+\n" + quals[i] + b"\n" for i in sel)
    bloom.close()
    store.close()


def test_bgzf_input_equals_plain_input(sdb, tmp_path):
    """a BGZF file (bgzip) is inflated block-parallel; table, totals and per-read outputs equal those of the plain file,
    also when ordinary gzip members follow the blocks"""
    from conftest import bgzf
    data = b"".join(_fastq_bytes(sdb, 30000, seed=61))
    more = b"".join(_fastq_bytes(sdb, 2000, seed=62))
    plain, packed, mixed = str(tmp_path / "a.fastq"), str(tmp_path / "a.fastq.gz"), str(tmp_path / "m.fastq.gz")
    open(plain, "wb").write(data)
    open(packed, "wb").write(bgzf(data))
    open(mixed, "wb").write(bgzf(data, eof_marker=False) + gzip.compress(more))
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    t0, _, tot0 = host.match_files(store, [plain], kraken_out_path=str(tmp_path / "k0"), taxids=sdb.taxids)
    t1, _, tot1 = host.match_files(store, [packed], kraken_out_path=str(tmp_path / "k1"), taxids=sdb.taxids)
    assert np.array_equal(t0, t1) and (tot0.reads, tot0.kmers, tot0.bps) == (tot1.reads, tot1.kmers, tot1.bps)
    assert open(str(tmp_path / "k0"), "rb").read() == open(str(tmp_path / "k1"), "rb").read()
    open(plain, "wb").write(data + more)
    t2, _, tot2 = host.match_files(store, [plain])
    t3, _, tot3 = host.match_files(store, [mixed])
    assert np.array_equal(t2, t3) and tot2.reads == tot3.reads == 32000
    bad = bytearray(bgzf(data))
    bad[len(bad) // 2] ^= 0x20
    open(packed, "wb").write(bytes(bad))
    with pytest.raises(RuntimeError):
        host.match_files(store, [packed])
    store.close()


def test_gzip_outputs_are_multi_member_and_round_trip(sdb, tmp_path):
    """.gz outputs are compressed by the formatting threads as BGZF blocks: the content must equal the plain outputs,
    and the library's own gzip reader must take the file back"""
    path = str(tmp_path / "in.fastq")
    open(path, "wb").write(b"".join(_fastq_bytes(sdb, 40000, seed=53)))
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    plain = dict(filtered_path=str(tmp_path / "f.fastq"), kraken_out_path=str(tmp_path / "k.out"))
    packed = dict(filtered_path=str(tmp_path / "f.fastq.gz"), kraken_out_path=str(tmp_path / "k.out.gz"))
    t1, _, tot1 = host.match_files(store, [path], taxids=sdb.taxids, write_all=True, **plain)
    t2, _, tot2 = host.match_files(store, [path], taxids=sdb.taxids, write_all=True, **packed)
    assert np.array_equal(t1, t2) and tot1.filtered_reads == tot2.filtered_reads > 1000
    for a, b in zip(plain.values(), packed.values()):
        raw = open(b, "rb").read()
        # BGZF blocks (what bgzip writes): block-parallel readers, this library's among them, can take the file apart
        assert raw.count(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00") > 10
        assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
        assert gzip.decompress(raw) == open(a, "rb").read()
        assert host.gunzip_parallel(raw, os.path.getsize(a), 4) == open(a, "rb").read()
    t3, _, tot3 = host.match_files(store, [packed["filtered_path"]])
    assert tot3.reads == tot1.filtered_reads
    # nothing to write: still a valid (empty) gzip file
    junk = str(tmp_path / "junk.fastq")
    open(junk, "wb").write(b"@r\n" + b"ACGT" * 20 + b"\n+\n" + b"I" * 80 + b"\n")
    none = str(tmp_path / "none.fastq.gz")
    host.match_files(store, [junk], filtered_path=none)
    assert gzip.decompress(open(none, "rb").read()) == b""
    store.close()


@pytest.mark.skipif(not os.path.exists("/dev/full"), reason="needs /dev/full")
def test_output_write_failure_is_reported(sdb, tmp_path):
    """the writers run on their own threads: a write that fails (disk full) must still fail the call"""
    path = str(tmp_path / "in.fastq")
    open(path, "wb").write(b"".join(_fastq_bytes(sdb, 5000, seed=47)))
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    with pytest.raises(RuntimeError, match="write"):
        host.match_files(store, [path], kraken_out_path="/dev/full", taxids=sdb.taxids, write_all=True)
    with pytest.raises(RuntimeError, match="write"):
        host.match_files(store, [path], filtered_path="/dev/full")
    t, _, tot = host.match_files(store, [path])  # the store and the library are fine afterwards
    assert tot.reads == 5000
    store.close()


@pytest.mark.parametrize("readers", [1, 3, 4, 7])
def test_match_files_few_blocks_many_readers(sdb, tmp_path, monkeypatch, readers):
    """files with fewer blocks than reader threads, processed repeatedly (pinned blocks come back from the pool): block i
    must always be the one the consumer gets for index i"""
    data = b"".join(_fastq_bytes(sdb, 600, seed=47))
    path = str(tmp_path / "few.fastq")
    open(path, "wb").write(data)
    want_t, want_tot = _oracle_file(sdb, data)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    monkeypatch.setenv("GS_HOST_READERS", str(readers))
    for block in (len(data) // 5 + 7, len(data) // 2 + 3, len(data) + 100):
        monkeypatch.setenv("GS_HOST_BLOCK_BYTES", str(block))
        for _ in range(3):
            t, _, tot = host.match_files(store, [path])
            assert (tot.reads, tot.kmers, tot.bps) == want_tot
            assert np.array_equal(t, want_t)
    store.close()


def test_refused_first_chunk_of_a_multi_batch_gz_while_its_upload_runs(sdb, tmp_path, monkeypatch):
    """a single-member .gz larger than its first batch is uploaded whole by a thread of the device decoder while the
    batches are decoded.  When the first chunk is refused (here: a record whose sequence runs over two lines) the job falls back and
    unmaps the file -- the decoder must have been parked BEFORE that, or the thread copies out of a mapping that is gone.  Result =
    the run through the host decoders."""
    import zlib
    n = 400_000
    seq, off = synth.reads_host(sdb.genomes, n, read_len=150, seed=41)
    a = np.empty((n, 2 + 8 + 1 + 150 + 3 + 150 + 1), dtype=np.uint8)
    a[:, 0], a[:, 1] = ord("@"), ord("r")
    idx = np.arange(n, dtype=np.int64)
    for d in range(8):
        a[:, 9 - d] = (idx % 10 + 48).astype(np.uint8)
        idx //= 10
    a[:, 10] = 10
    a[:, 11:161] = seq.reshape(n, 150)
    a[:, 161:164] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    a[:, 164:314] = ord("I")
    a[:, 314] = 10
    first = b"@first\n" + seq[:75].tobytes() + b"\n" + seq[75:150].tobytes() + b"\n+\n" + b"I" * 150 + b"\n"  # sequence over two lines
    text = first + a.reshape(-1).tobytes()
    z = zlib.compressobj(1, zlib.DEFLATED, 31)
    path = str(tmp_path / "big.fastq.gz")
    with open(path, "wb") as f:
        f.write(z.compress(text) + z.flush())
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    monkeypatch.setenv("GS_GUNZIP_SLOTS", "16")        # 1 MiB of compressed data per batch: the upload outlives the first one
    monkeypatch.setenv("GS_HOST_GUNZIP_FIRST", "0")
    table, _, tot = host.match_files(store, [path])
    monkeypatch.setenv("GS_DEVICE_INFLATE", "0")
    want, _, tot2 = host.match_files(store, [path])
    assert tot.reads == n + 1 == tot2.reads and np.array_equal(table, want)
    # and the same file without the odd first record goes through the device decoder batch by batch
    monkeypatch.delenv("GS_DEVICE_INFLATE")
    z = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(path, "wb") as f:
        f.write(z.compress(text[len(first):]) + z.flush())
    table2, _, tot3 = host.match_files(store, [path])
    orun = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi))
    orun.submit(seq, off, threads=8, per_read=False)
    otable, _ = orun.finish()
    assert tot3.reads == n and np.array_equal(table2, otable)
    store.close()


@pytest.mark.parametrize("with_probs", [False, True])
@pytest.mark.parametrize("container", ["bgzf", "gzip", "plain"])
def test_filter_outputs_written_from_the_device(sdb, tmp_path, monkeypatch, container, with_probs):
    """The reference's default shape of the filter goal -- gzip in, gzip out (gzipFastqOutput, C/GSConfigKey.java:155;
    C/goals/FilterGoal.java:76) -- with the output side on the device: accepted and dumped records gathered there
    (gs_filter_compact_text), compressed there (gs_deflater_pack), only the files' bytes cross PCIe.  The .gz files must hold exactly
    what the plain files of the host formatter hold (GS_DEVICE_OUTPUT=0), and must be BGZF that the device inflater takes back."""
    from conftest import bgzf
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:4])]
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys)
    gb = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
    recs = _fastq_bytes(sdb, 6000, seed=77)
    if with_probs:  # quality lines that differ from record to record
        rng = np.random.default_rng(4)
        for i, r in enumerate(recs):
            parts = r.split(b"\n")
            parts[3] = bytes(rng.integers(35, 74, len(parts[3]), dtype=np.uint8))
            recs[i] = b"\n".join(parts)
    data = b"".join(recs) + b"@last\nACGT\n+\nIII"  # (an unterminated tail: goes through the reference-exact parser, behind the device's chunks)
    path = str(tmp_path / ("in.fastq" if container == "plain" else "in.fastq.gz"))
    if container == "bgzf":
        open(path, "wb").write(bgzf(data, block=30000, level=1))
    elif container == "gzip":
        open(path, "wb").write(gzip.compress(data, compresslevel=1, mtime=0))
    else:
        open(path, "wb").write(data)
    monkeypatch.setenv("GS_HOST_BGZF_TEXT", "400000")  # several feeds
    monkeypatch.setenv("GS_HOST_BLOCK_BYTES", "300000")
    monkeypatch.setenv("GS_DEVICE_OUTPUT", "0")
    a0, r0 = str(tmp_path / "acc0.fastq"), str(tmp_path / "rest0.fastq")
    tot0 = host.filter_files(gb, 31, [path], filtered_path=a0, rest_path=r0, with_probs=with_probs)
    monkeypatch.setenv("GS_DEVICE_OUTPUT", "1")
    a1, r1 = str(tmp_path / "acc1.fastq.gz"), str(tmp_path / "rest1.fastq.gz")
    tot1 = host.filter_files(gb, 31, [path], filtered_path=a1, rest_path=r1, with_probs=with_probs)
    assert (tot1.reads, tot1.filtered_reads) == (tot0.reads, tot0.filtered_reads) and tot0.reads >= 6000 and 0 < tot0.filtered_reads < 6000
    for gz_p, plain_p in ((a1, a0), (r1, r0)):
        comp, want = open(gz_p, "rb").read(), open(plain_p, "rb").read()
        assert gzip.decompress(comp) == want
        members, reached = ga.bgzf_members(comp)
        assert reached == len(comp) and members[-1][2] == 0  # BGZF to the end, closed by the empty member
        text, status = ga.inflate_members(comp, [m for m in members if m[2]])
        assert not status.any() and text.tobytes() == want
    # ... and plain outputs through the device path (BGZF / gzip input: the text never comes to the host)
    if container != "plain":
        a2, r2 = str(tmp_path / "acc2.fastq"), str(tmp_path / "rest2.fastq")
        host.filter_files(gb, 31, [path], filtered_path=a2, rest_path=r2, with_probs=with_probs)
        assert open(a2, "rb").read() == open(a0, "rb").read() and open(r2, "rb").read() == open(r0, "rb").read()
    gb.close()


@pytest.mark.parametrize("container", ["bgzf", "gzip", "plain"])
def test_match_filtered_fastq_written_from_the_device(sdb, tmp_path, monkeypatch, container):
    """writeFilteredFastq with gzipFastqOutput (C/goals/MatchResultGoal.java:106, C/GSConfigKey.java:155, :308) and no Kraken-style
    lines: the reads matchRead returned true for are gathered and compressed on the device (gs_match_compact_text,
    gs_deflater_pack); file, table and totals equal the host formatter's (GS_DEVICE_OUTPUT=0)"""
    from conftest import bgzf
    recs = _fastq_bytes(sdb, 5000, seed=91)
    data = b"".join(recs)
    path = str(tmp_path / ("in.fastq" if container == "plain" else "in.fastq.gz"))
    if container == "bgzf":
        open(path, "wb").write(bgzf(data, block=30000, level=1))
    elif container == "gzip":
        open(path, "wb").write(gzip.compress(data, compresslevel=1, mtime=0))
    else:
        open(path, "wb").write(data)
    monkeypatch.setenv("GS_HOST_BGZF_TEXT", "300000")
    monkeypatch.setenv("GS_HOST_BLOCK_BYTES", "200000")
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    res = {}
    for dev in ("0", "1"):
        monkeypatch.setenv("GS_DEVICE_OUTPUT", dev)
        f = str(tmp_path / f"flt{dev}.fastq.gz")
        table, _, tot = host.match_files(store, [path], filtered_path=f)
        res[dev] = (table, tot.reads, tot.filtered_reads, gzip.decompress(open(f, "rb").read()))
    assert np.array_equal(res["0"][0], res["1"][0]) and res["0"][1:3] == res["1"][1:3] and 0 < res["1"][2] < 5000
    assert res["0"][3] == res["1"][3] and res["1"][3].count(b"\n") == 4 * res["1"][2]
    if container != "plain":  # a plain filtered file from compressed input takes the device path as well
        f = str(tmp_path / "flt2.fastq")
        host.match_files(store, [path], filtered_path=f)
        assert open(f, "rb").read() == res["0"][3]
    store.close()


@pytest.mark.parametrize("container", ["plain", "gzip", "bgzf"])
def test_device_written_gz_keeps_the_order_around_host_formatted_stretches(sdb, tmp_path, monkeypatch, container):
    """Chunks gathered on the device wait there to be compressed together (gs_deflater_append); a stretch of the file that the device
    refuses (a record with its sequence over two lines) is formatted on the host and written through the same file object.  The .gz
    output must hold the records in input order -- what waits on the device goes out before any host-side write -- and equal the
    plain output of the host formatter, for the accepted and for the dumped reads."""
    from conftest import bgzf
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:4])]
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys)
    gb = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
    recs = [r for r in _fastq_bytes(sdb, 9000, seed=301) if len(r) > 20]
    seq, off = synth.reads_host(sdb.genomes, 3, read_len=150, seed=5)
    two_lines = [b"@ml%d\n%s\n%s\n+\n%s\n" % (i, seq[int(off[i]):int(off[i]) + 70].tobytes(), seq[int(off[i]) + 70:int(off[i + 1])].tobytes(), b"F" * 150) for i in range(3)]
    data = b"".join(recs[:2500]) + two_lines[0] + b"".join(recs[2500:5200]) + two_lines[1] + two_lines[2] + b"".join(recs[5200:])
    path = str(tmp_path / ("in.fastq" if container == "plain" else "in.fastq.gz"))
    if container == "bgzf":
        open(path, "wb").write(bgzf(data, block=30000, level=1))
    elif container == "gzip":
        open(path, "wb").write(gzip.compress(data, compresslevel=1, mtime=0))
    else:
        open(path, "wb").write(data)
    monkeypatch.setenv("GS_HOST_BGZF_TEXT", "300000")
    monkeypatch.setenv("GS_HOST_BLOCK_BYTES", "200000")
    monkeypatch.setenv("GS_DEVICE_OUTPUT", "0")
    a0, r0 = str(tmp_path / "acc0.fastq"), str(tmp_path / "rest0.fastq")
    tot0 = host.filter_files(gb, 31, [path, path], filtered_path=a0, rest_path=r0)
    monkeypatch.setenv("GS_DEVICE_OUTPUT", "1")
    a1, r1 = str(tmp_path / "acc1.fastq.gz"), str(tmp_path / "rest1.fastq.gz")
    tot1 = host.filter_files(gb, 31, [path, path], filtered_path=a1, rest_path=r1)
    assert (tot1.reads, tot1.filtered_reads) == (tot0.reads, tot0.filtered_reads) and tot0.reads == 2 * (len(recs) + 3)
    want_a, want_r = open(a0, "rb").read(), open(r0, "rb").read()
    assert gzip.decompress(open(a1, "rb").read()) == want_a and gzip.decompress(open(r1, "rb").read()) == want_r
    assert want_a.count(b"@ml") + want_r.count(b"@ml") == 6  # (the two-line records are in the outputs, rewritten in four lines)
    gb.close()
