"""kernel time of gs_match_kernel per read stream (developer tool): the bench stream (half of the reads from the store), a
hit-only stream (every read from the store, 1 % substitutions) and a miss-only stream (reads from other genomes), on the
config-2 store or, with --large, the 47 M-k-mer store.  With library paths as arguments every library (a build variant,
tools/ablate.sh) is timed in a child process:   python tools/stream_times.py [--large] [lib.so ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
large = "--large" in sys.argv
only = next((a.split("=")[1] for a in sys.argv if a.startswith("--stream=")), None)  # bench | hit | miss (for profiling)
if args and os.environ.get("GS_STREAM_CHILD") != "1":
    for lib in args:
        env = dict(os.environ, GS_LIBGSGPU=os.path.abspath(lib), GS_STREAM_CHILD="1")
        out = subprocess.run([sys.executable, os.path.abspath(__file__)] + (["--large"] if large else []), env=env,
                             capture_output=True, text=True)
        print(f"{os.path.basename(lib):24s} {out.stdout.strip() or out.stderr.strip()[-300:]}", flush=True)
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

n = 10_000_000
db = synth.SynthDB(genera=25, species_per_genus=20) if large else synth.SynthDB()
other = synth.SynthDB(seed=43)
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
dseq = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
doff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))


def timed():
    for _ in range(2):
        m.reset()
        m.submit(dseq, doff, 0, n_reads=n)
    m.sync()
    l0, t0 = m.kernel_time()
    for _ in range(4):
        m.reset()
        m.submit(dseq, doff, 0, n_reads=n)
    m.sync()
    l1, t1 = m.kernel_time()
    return (t1 - t0) / (l1 - l0)


res = []
gen = torch.from_numpy(db.genomes).cuda()
synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff)
if only in (None, "bench"):
    res.append(("bench", timed()))
# hit-only: the reads of the bench stream that come from the store (gs_synth.hip: bit 0 of sy_hash(seed, read, 0) = background)
seq, off = synth.reads_host(db.genomes, 200_000)
with np.errstate(over="ignore"):
    z = np.uint64(4242) + np.arange(200_000, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
hit_rows = np.flatnonzero((z & np.uint64(1)) == 0)[:65536]
hseq = torch.from_numpy(seq.reshape(-1, 150)[hit_rows].copy()).cuda()
dseq.view(-1, 150)[:] = hseq.repeat((n + len(hit_rows) - 1) // len(hit_rows), 1)[:n]
if only in (None, "hit"):
    res.append(("hit-only", timed()))
gen2 = torch.from_numpy(other.genomes).cuda()
synth.reads_device(gen2, other.genomes.shape[0], other.genomes.shape[1], n, dseq, doff)
if only in (None, "miss"):
    res.append(("miss-only", timed()))
print("  ".join(f"{k} {v:7.3f} ms" for k, v in res))
