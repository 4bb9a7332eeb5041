"""fixed cost of a gs_host_match_files call on small files (developer tool)"""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import genestrip_amd as ga
from genestrip_amd import host, synth
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
seq, off = synth.reads_host(db.genomes, 100000)
blk = seq.tobytes()
d = tempfile.mkdtemp()
for n in (100, 10000, 100000):
    p = os.path.join(d, f"in{n}.fastq")
    open(p, "wb").write(b"".join(b"@r%d\n" % i + blk[i*150:(i+1)*150] + b"\n+\n" + b"I"*150 + b"\n" for i in range(n)))
    for rep in range(3):
        t0 = time.perf_counter()
        _, _, tot = host.match_files(store, [p])
        print(f"{n} reads: {1e3 * (time.perf_counter() - t0):.1f} ms (parse {tot.seconds_parse*1e3:.1f} total {tot.seconds_total*1e3:.1f})", flush=True)
    t0 = time.perf_counter()
    host.match_files(store, [p] * 20)
    print(f"20 x {n} reads in one call: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
