"""tools/check_passes.py -- one-off parity check of the memory-planned, overlapped pass pipeline at a size the
test suite does not reach (3 M pairs, ~10 passes forced by a small HBM budget)."""
import sys, time
sys.path.insert(0, ".")
from superplus_amd import synth
from oracle import pyoracle
from tests import util

genome = synth.make_genome(20_000_000, 15, repeat_frac=0.02)
rs = synth.make_reads(genome, 3_000_000, 16).numpy()
t = time.time()
ref, d = util.run_both(pyoracle, rs, K=48)
st = util.check_parity(ref, d)
print("one pass: parity ok, peak %.2f GB, %.1f s" % (st["hbm_bytes_peak"] / 1e9, time.time() - t))
budget = st["hbm_bytes_peak"] - int(0.55 * 32 * st["n_records"])
del d
from superplus_amd.dfk import Dfk
d = Dfk(K=48, keep_pre_adjacency=True, hbm_budget_bytes=budget)
d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
st = util.check_parity(ref, d)
print("budget %.2f GB: parity ok, %d passes" % (budget / 1e9, st["n_passes"]))
