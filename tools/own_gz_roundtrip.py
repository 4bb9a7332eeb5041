"""reads the library's own multi-member .gz output back through the file pipeline (developer tool)"""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import genestrip_amd as ga
from genestrip_amd import host, synth
n = 4_000_000
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
seq, off = synth.reads_host(db.genomes, n)
blk = seq.tobytes()
d = tempfile.mkdtemp()
p = os.path.join(d, "in.fastq")
open(p, "wb").write(b"".join(b"@r%d\n" % i + blk[i*150:(i+1)*150] + b"\n+\n" + b"I"*150 + b"\n" for i in range(n)))
f = os.path.join(d, "f.fastq.gz")
_, _, tot = host.match_files(store, [p], filtered_path=f, with_probs=True)
print("filtered reads", tot.filtered_reads, "gz size", os.path.getsize(f) / 1e6, "MB", flush=True)
for _ in range(2):
    t0 = time.perf_counter()
    _, _, t2 = host.match_files(store, [f])
    dt = time.perf_counter() - t0
    print(f"own multi-member output read back: {dt:.2f} s -> {t2.reads * 150 / dt / 1e9:.2f} Gbp/s, reads {t2.reads}", flush=True)
