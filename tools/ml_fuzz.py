"""campaign over the device record search for general FASTQ (developer tool): random multi-line files of tests/test_gpu_fastq_ml.py's
generator under many seeds, chunked at random piece sizes through gs_match_submit_fastq_ml, tables against the oracle.
python tools/ml_fuzz.py [first_seed] [n_seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402
from oracle import gs_oracle as orc  # noqa: E402
import test_gpu_fastq_ml as T  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sdb = synth.SynthDB(k=31, genera=2, species_per_genus=3, genome_len=20000, seed=33)
store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
m = ga.FastqKMerMatcher(store)
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    data = T._multiline_fastq(sdb, int(rng.integers(200, 2500)), seed=seed, max_seq_lines=int(rng.integers(2, 9)))
    want, rd = T._oracle_table(sdb, data)
    piece = int(rng.choice([700, 4093, 30011, 1 << 20]))
    m.reset()
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
            if carry:
                tail = orc.parse_fastq(carry, fasta=False, k=31)
                if tail["n_reads"]:
                    m.submit(tail["seq"], tail["seq_off"], done, n_reads=tail["n_reads"])
                    done += tail["n_reads"]
            break
    m.sync()
    table = m.finish()[0]
    ok = done == rd["n_reads"] and np.array_equal(table, want) and m.text_status()[0] < 0
    bad += not ok
    if not ok:
        print("seed %d piece %d: MISMATCH (records %d / %d)" % (seed, piece, done, rd["n_reads"]), flush=True)
print("%d seeds, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
