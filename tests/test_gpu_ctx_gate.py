"""The context-keyed minimizer gate of big stores (gs_layout.h gs_gate_ctx_key, GsDbDev::mgate_ctx): forced onto small stores
(GS_GATE_CTX_MIN_DISTINCT=1) it must leave every result as it is -- the gate has no false negatives whatever it is keyed by --,
survive a save / load round trip and a striped build, and stay off below k = 22.  Run with -m gpu."""
import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import synth
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu


def _check(store, db, k, seq, off, **cfg):
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(**cfg))
    cv, fl = m.match_reads(seq, off)
    t, _ = m.finish()
    m.close()
    orun = orc.MatchRun(orc.DB(k, db.kmers, db.value_idx, db.n_values, db.parent_vi), **cfg)
    ocv, ofl = orun.submit(seq, off)
    ot, _ = orun.finish()
    assert np.array_equal(t, ot) and np.array_equal(cv, ocv) and np.array_equal(fl, ofl)


@pytest.mark.parametrize("k", [31, 27, 22, 21])
def test_context_keyed_gate_changes_no_result(k, monkeypatch, tmp_path):
    db = synth.SynthDB(k=k, genera=3, species_per_genus=3, genome_len=20000, seed=11)
    monkeypatch.setenv("GS_GATE_CTX_MIN_DISTINCT", "1")
    store = ga.DeviceKMerStore(k, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    other = synth.SynthDB(k=k, genera=2, species_per_genus=2, genome_len=20000, seed=99)
    seq, off = synth.reads_host(db.genomes, 12000, read_len=150, seed=5)
    seq = seq.copy()
    seq[7::4001] = ord("N")
    seq2, off2 = synth.reads_host(other.genomes, 12000, read_len=150, seed=6)
    long_seq, long_off = synth.reads_host(db.genomes, 600, read_len=700, seed=8)
    _check(store, db, k, seq, off)
    _check(store, db, k, seq2, off2)
    _check(store, db, k, long_seq, long_off)          # the long-read kernel asks the store for the mode
    _check(store, db, k, seq, off, max_paths=128)     # so do the 128-path kernels
    path = tmp_path / "ctx.gss"
    store.save(path)
    store.close()
    # the mode travels in bit 31 of the header word behind magic (8) | gs_db_info (96) | bucket_bits (4); it stays off below k = 22
    flag = int.from_bytes(path.read_bytes()[108:112], "little") >> 31
    assert flag == (1 if k >= 22 else 0)
    again = ga.DeviceKMerStore.load(path)
    _check(again, db, k, seq, off)
    _check(again, db, k, seq2, off2)
    again.close()
    stripes = ga.DeviceKMerStore.striped(k, db.kmers, db.value_idx, db.n_values, db.parent_vi, devices=(0, 0, 0))
    _check(stripes[1], db, k, seq, off)
    _check(stripes[2], db, k, seq2, off2)
    for s in stripes:
        s.close()
