"""end-to-end rate of gs_host_match_files over several gzip FASTQ files, read side by side vs one after the other
(developer tool).  argv[1] = number of files (default 8), argv[2] = reads per file (default 1 M)."""
import gzip
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import host, synth  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
d = tempfile.mkdtemp(prefix="gsmulti")
paths = []
for f in range(nf):
    seq, off = synth.reads_host(db.genomes, n, seed=100 + f)
    L = 150
    qual = (b"FFFFFFFFFF:FFFFF,FFFFFFFFFFFFFF" * 5)[:L]
    blk = seq.tobytes()
    data = b"".join(b"@A00123:45:HXX:1:1101:%d:%d 1:N:0:ACGT\n" % (f, i) + blk[i * L:(i + 1) * L] + b"\n+\n" + qual + b"\n" for i in range(n))
    p = os.path.join(d, f"lane{f}.fastq.gz")
    with gzip.open(p, "wb", compresslevel=4) as g:
        g.write(data)
    paths.append(p)
    print(f"wrote {p}: {len(data) / 1e6:.0f} MB text, {os.path.getsize(p) / 1e6:.0f} MB gzip", flush=True)
for par in ("1", "0"):
    os.environ["GS_HOST_PARALLEL_FILES"] = par
    t0 = time.perf_counter()
    table, _, tot = host.match_files(store, paths)
    dt = time.perf_counter() - t0
    print(f"GS_HOST_PARALLEL_FILES={par}: {dt:.2f} s -> {nf * n * 150 / dt / 1e9:.2f} Gbp/s, reads {tot.reads}, "
          f"table checksum {int(table.sum())}", flush=True)
shutil.rmtree(d)
