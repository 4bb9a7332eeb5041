"""CPU-side checks of the drop-in boundary: libgsgpu.so loads, exports every symbol include/gsgpu.h declares,
and refuses to compute without a GPU (there is no CPU fallback in the product path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import genestrip_amd as ga
from conftest import ROOT


def test_library_loads_and_exports_every_declared_symbol():
    L = ga.lib()
    header = open(os.path.join(ROOT, "include", "gsgpu.h")).read()
    declared = set(re.findall(r"\b(gs_[a-z_0-9]+)\s*\(", header))
    assert declared == set(ga.ABI_SYMBOLS), declared ^ set(ga.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert ga.abi_version() == 3
    assert L.gs_strerror(-6).decode() == "no usable gfx950 device"


def test_device_cache_trim_needs_no_device():
    """gs_device_cache_trim frees the device blocks that wait for the next run; with nothing waiting (no GPU here) it just returns"""
    assert ga.lib().gs_device_cache_trim() == 0


def test_host_layer_exports_every_declared_symbol():
    from genestrip_amd import host
    L = host.lib()
    header = open(os.path.join(ROOT, "include", "gshost.h")).read()
    declared = set(re.findall(r"\b(gs_(?:fastq|host)_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 9
    for name in declared:
        assert hasattr(L, name), name


def test_integration_md_names_every_entry_point():
    """INTEGRATION.md maps each entry point of both headers to the reference interface it replaces (or says that there is
    none): by its full name, or as a `_suffix` inside a `gs_family_a / _b / _c` group"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = set()
    for h in ("gsgpu.h", "gshost.h"):
        names |= set(re.findall(r"\b(gs_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", h)).read()))
    missing = [n for n in sorted(names) if n not in doc and ("_" + n.rsplit("_", 1)[1]) not in doc]
    assert not missing, missing


def test_header_enums_match_binding():
    header = open(os.path.join(ROOT, "include", "gsgpu.h")).read()
    cols = re.search(r"enum \{\s*GS_C_READS = 0,(.*?)GS_N_COLS", header, re.S).group(1)
    assert len(re.findall(r"GS_C_", cols)) + 1 == ga.N_COLS
    sums = re.search(r"enum \{ GS_S_READS = 0,(.*?)GS_N_SUMS", header, re.S).group(1)
    assert len(re.findall(r"GS_S_", sums)) + 1 == ga.N_SUMS


@pytest.mark.skipif(ga.device_count() > 0, reason="only meaningful without a GPU")
def test_fails_loudly_without_gpu():
    with pytest.raises(ga.GsError) as e:
        ga.DeviceKMerStore(2, [5], [0], 1)
    assert e.value.code == -6
    with pytest.raises(ga.GsError):
        ga.DeviceBloomFilter(ga.BLOOM_XOR, 64, [1], np.zeros(1, np.uint64))


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "genestrip_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "gs_oracle" not in txt and "libgsoracle" not in txt, f


def test_synth_reads_deterministic_and_shaped():
    from genestrip_amd import synth
    db = synth.SynthDB(k=31, genera=2, species_per_genus=2, genome_len=5000, seed=7)
    assert np.all(np.diff(db.kmers) > 0) and db.n_values == 7
    seq, off = synth.reads_host(db.genomes, 1000, read_len=150, seed=3)
    seq2, _ = synth.reads_host(db.genomes, 400, read_len=150, seed=3, first=600)
    assert np.array_equal(seq[600 * 150:], seq2)
    assert set(np.unique(seq).tolist()) <= set(b"ACGTN")
    assert off[-1] == 150000
