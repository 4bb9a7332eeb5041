import os, sys
sys.path.insert(0, os.getcwd())
import genestrip_amd as ga
from genestrip_amd import host, synth
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
n = 100000
seq, off = synth.reads_host(db.genomes, n)
L = 150
blk = seq.tobytes()
path = "/tmp/dbg.fastq"
open(path, "wb").write(b"".join(b"@r%d\n" % i + blk[i*L:(i+1)*L] + b"\n+\n" + b"I"*L + b"\n" for i in range(n)))
for i in range(4):
    t, _, tot = host.match_files(store, [path])
    print(i, tot.reads, tot.kmers, int(t[:, 0].sum()), flush=True)
os.environ["GS_HOST_FAST"] = "0"
t, _, tot = host.match_files(store, [path])
print("slow", tot.reads, tot.kmers, int(t[:, 0].sum()))
