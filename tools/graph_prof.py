"""tools/graph_prof.py [GENOME_MB] [PAIRS] -- count a synthetic set, then build the unipath graph (row f-1); run under
rocprofv3 --kernel-trace --stats to see where the device half goes."""
import sys, time, torch
sys.path.insert(0, '.')
from superplus_amd import synth
from superplus_amd.dfk import Dfk
G = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 1_550_000_000
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 450_000_000
dev = torch.device("cuda:0")
genome = synth.make_genome(G, 20250, device=dev)
rs = synth.make_reads(genome, pairs, 20267)
del genome
torch.cuda.synchronize(); torch.cuda.empty_cache()
d = Dfk(K=48, device=0)
d.count_device(rs.packed, rs.base_off, rs.read_len, rs.pq_bytes, rs.pq_off, rs.bc)
st = d.stats(); print("solid", st["n_solid"], "ms", st["ms_total"], flush=True)
del rs; torch.cuda.empty_cache()
t0 = time.time(); g = d.graph_build(); t1 = time.time()
print("graph_build %.2f s: %s" % (t1 - t0, g), flush=True)
