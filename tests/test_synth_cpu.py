"""The synthetic-input builders (genestrip_amd/synth.py, libgssynth.so) against the oracle's restatement: the index
filter a bench / test workload is given must be the one BloomIndexGoal would have built.  CPU only."""
import numpy as np
import pytest

from genestrip_amd import synth
from oracle import gs_oracle as orc


def test_java_random_longs_are_the_reference_hash_factors():
    assert synth.java_random_longs(42, 3).tolist() == [-5025562857975149833, -5843495416241995736, 5694868678511409995]


@pytest.mark.parametrize("n,fpp", [(1000, 1e-8), (50_000, 1e-8), (47_038_401, 1e-8), (20_000, 0.05), (7, 0.5)])
def test_xor_bloom_geometry_equals_the_oracle(n, fpp):
    ob = orc.Bloom(orc.BLOOM_XOR, n, fpp)
    bits, hashes, factors = synth.xor_bloom_geometry(n, fpp)
    assert (bits, hashes) == (ob.bits, ob.hashes)
    assert np.array_equal(factors, ob.hash_factors)
    if fpp == 1e-8:
        assert hashes == 27


def test_xor_bloom_words_equal_the_oracle():
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 1 << 62, 30_000, dtype=np.int64)
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys)
    bits, hashes, factors = synth.xor_bloom_geometry(len(keys), 1e-8)
    words = synth.xor_bloom_host(keys, bits, factors)
    assert np.array_equal(words, ob.words)
