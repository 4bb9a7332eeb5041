"""host builder phases without a GPU (developer tool):  GS_BUILD_THREADS=8 python tools/build_dry.py [genera]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GS_BUILD_TRACE"] = "1"
os.environ["GS_BUILD_DRYRUN"] = "1"
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

for genera in [int(x) for x in sys.argv[1:]] or [25]:
    db = synth.SynthDB(genera=genera, species_per_genus=20)
    t0 = time.time()
    try:
        ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    except ga.GsError as e:
        print(e)
    print("genera %d: %d k-mers, layout %.2f s" % (genera, db.n_entries, time.time() - t0), flush=True)
