"""Text mode (gs_match_submit_text): raw FASTQ chunks whose records are found on the device must give exactly what the
reference parser + matchRead give (oracle: gs_oracle's restatement of AbstractFastqReader.doReadFastq), and chunks that
are not plain four-line FASTQ must be refused without touching the run's state.  Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import synth
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(genera=2, species_per_genus=3, genome_len=20_000, seed=7)


@pytest.fixture(scope="module")
def store(sdb):
    s = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    yield s
    s.close()


def _records(sdb, n, seed=3, crlf=False):
    """n FASTQ records (descriptor, read, quality) with the odd shapes the parser has to keep: N bases, reads shorter
    than k, empty reads, reads long enough for the long-read kernel, qualities longer than the read"""
    rng = np.random.default_rng(seed)
    seq, off = synth.reads_host(sdb.genomes, n, seed=seed)
    recs = []
    for i in range(n):
        r = bytes(seq[int(off[i]):int(off[i + 1])])
        kind = i % 23
        if kind == 3:
            r = r[:rng.integers(0, 31)]                      # shorter than k, possibly empty
        elif kind == 5:
            r = r + bytes(seq[int(off[(i + 1) % n]):int(off[(i + 1) % n + 1])]) * 2   # 450 bases: long-read kernel
        elif kind == 7:
            p = int(rng.integers(0, len(r)))
            r = r[:p] + b"N" + r[p + 1:]
        elif kind == 11:
            r = r.lower()                                    # lower case is invalid in the reference
        q = b"I" * len(r) + (b"#" * 5 if kind == 13 else b"")
        d = b"@read%d some text" % i if kind != 17 else b""   # an empty descriptor line is a descriptor
        if kind == 19:
            r = b"+" + r[1:]                                  # a FIRST sequence line may start with '+'
        recs.append((d, r, q))
    return recs


def _text(recs, crlf=False):
    nl = b"\r\n" if crlf else b"\n"
    return b"".join(d + nl + r + nl + b"+" + nl + q + nl for d, r, q in recs)


def _oracle_on_text(sdb, text, first_read_no=0, **cfg):
    """the reference path: parse the bytes (restated parser), then matchRead"""
    p = orc.parse_fastq(text, k=31)
    seq, off = p["seq"], p["seq_off"]
    if len(seq) == 0:
        seq = np.zeros(1, dtype=np.uint8)
    odb = orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    run = orc.MatchRun(odb, **cfg)
    cv, fl = run.submit(seq, off, first_read_no)
    t, _ = run.finish()
    return t, cv, fl, (int(p["n_reads"]), int(p["total_kmers"]), int(p["total_bps"]))


@pytest.mark.parametrize("crlf", [False, True])
def test_text_chunks_equal_reference_parse(sdb, store, crlf):
    recs = _records(sdb, 4000)
    text = _text(recs, crlf)
    want_t, want_cv, want_fl, want_tot = _oracle_on_text(sdb, text)
    m = ga.FastqKMerMatcher(store)
    # three chunks cut at record boundaries, read numbers running on
    cuts = [0, 1300, 1301, 4000]
    got_cv, got_fl = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        chunk = np.frombuffer(_text(recs[a:b], crlf), dtype=np.uint8)
        cv = np.full(b - a, -7, dtype=np.int32)
        fl = np.full(b - a, 99, dtype=np.uint8)
        t = m.submit_text(chunk, first_read_no=a, class_vi=cv, flags=fl)
        m.text_wait_copy(t)
        m.sync()
        got_cv.append(cv)
        got_fl.append(fl)
    failed, bad, tot = m.text_status()
    assert failed == -1 and bad == -1
    assert tot == want_tot
    got_t, _ = m.finish()
    m.close()
    assert np.array_equal(got_t, want_t), np.argwhere(got_t != want_t)[:8]
    assert np.array_equal(np.concatenate(got_cv), want_cv)
    assert np.array_equal(np.concatenate(got_fl), want_fl)


def test_text_device_resident_chunk(sdb, store):
    import torch
    recs = _records(sdb, 1500, seed=9)
    text = _text(recs)
    want_t, want_cv, want_fl, want_tot = _oracle_on_text(sdb, text)
    m = ga.FastqKMerMatcher(store)
    d = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).cuda()
    cv = torch.full((1500,), -7, dtype=torch.int32, device="cuda")
    fl = torch.zeros(1500, dtype=torch.uint8, device="cuda")
    m.submit_text(d, n_lines=6000, class_vi=cv, flags=fl)
    m.sync()
    assert m.text_status() == (-1, -1, want_tot)
    got_t, _ = m.finish()
    m.close()
    assert np.array_equal(got_t, want_t)
    assert np.array_equal(cv.cpu().numpy(), want_cv)
    assert np.array_equal(fl.cpu().numpy(), want_fl)


def _bad_chunks(recs):
    d, r, q = recs[5]
    multi = _text(recs[:5]) + d + b"\n" + r[:70] + b"\n" + r[70:] + b"\n+\n" + q + b"\n" + _text(recs[6:9])  # 41 lines
    multi += d + b"\n" + r + b"\n+\n"  # pad to a multiple of four lines: 44
    shortq = _text(recs[:3]) + d + b"\n" + r + b"\n+\n" + q[:-1] + b"\n"
    nul = _text(recs[:2]) + d + b"\n" + r[:10] + b"\0" + r[11:] + b"\n+\n" + q + b"\n"
    noplus = _text(recs[:1]) + d + b"\n" + r + b"\n-\n" + q + b"\n"
    return {"multi-line sequence": multi, "short quality": shortq, "NUL byte": nul, "no plus line": noplus}


@pytest.mark.parametrize("what", ["multi-line sequence", "short quality", "NUL byte", "no plus line"])
def test_text_refuses_what_is_not_four_line_fastq(sdb, store, what):
    recs = [x for x in _records(sdb, 40, seed=5) if len(x[1]) > 100][:12]
    good = _text(recs)
    bad = _bad_chunks(recs)[what]
    assert bad.count(b"\n") % 4 == 0
    want_t, _, _, want_tot = _oracle_on_text(sdb, good)
    m = ga.FastqKMerMatcher(store)
    t0 = m.submit_text(good)
    t1 = m.submit_text(bad, first_read_no=12)
    t2 = m.submit_text(good, first_read_no=100)  # after a refusal every later chunk is skipped as well
    failed, first_bad, tot = m.text_status()
    assert (t0, t1, t2) == (0, 1, 2)
    assert failed == 1
    assert tot == want_tot
    if what in ("multi-line sequence", "short quality", "no plus line"):
        assert first_bad == {"multi-line sequence": 5, "short quality": 3, "no plus line": 1}[what]
    m.text_clear_error()
    assert m.text_status()[0] == -1
    got_t, _ = m.finish()
    assert np.array_equal(got_t, want_t)  # only the first chunk was counted
    m.reset()
    assert m.text_status() == (-1, -1, (0, 0, 0))
    m.close()


def test_text_argument_errors(store):
    m = ga.FastqKMerMatcher(store)
    with pytest.raises(ga.GsError):
        m.submit_text(b"@a\nACGT\n+\nIIII\n", n_lines=3)
    with pytest.raises(ga.GsError):
        m.text_wait_copy(5)
    # a newline count that does not match the text is refused on the device, not trusted
    m.submit_text(b"@a\nACGT\n+\nIIII\n@b\nACGT\n+\nIIII\n", n_lines=4)
    assert m.text_status()[0] == 0
    m.close()


def test_text_empty_chunk(store):
    m = ga.FastqKMerMatcher(store)
    m.submit_text(np.zeros(0, dtype=np.uint8), n_lines=0)
    assert m.text_status() == (-1, -1, (0, 0, 0))
    m.close()
