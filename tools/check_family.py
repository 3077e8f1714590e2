"""tools/check_family.py [LIB] -- one-off parity check on a repeat-family set too large for the test suite."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from superplus_amd import synth
from oracle import pyoracle
from tests import util

genome = synth.make_genome(20_000_000, 5, family_copies=6000)
rs = synth.make_reads(genome, 3_000_000, 6).numpy()
t = time.time()
ref, d = util.run_both(pyoracle, rs, K=48)
print("oracle+gpu %.1f s" % (time.time() - t), "solid", ref["n_solid"], d.solid_count(), "distinct", ref["n_distinct"], d.stats()["n_distinct"],
      "overflow items", d.stats()["n_overflow_items"])
util.check_parity(ref, d)
print("parity ok")
