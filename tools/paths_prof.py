"""tools/paths_prof.py [GENOME_MB] [PAIRS] -- count a synthetic set, build the graph, path the reads, index and mark
duplicates (rows f-1, f-2, f-4) on device-resident reads; run under rocprofv3 --kernel-trace --stats to see where the time goes."""
import sys, time, tempfile, torch
sys.path.insert(0, '.')
from superplus_amd import synth
from superplus_amd.dfk import Dfk
G = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 200_000_000
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 58_000_000
dev = torch.device("cuda:0")
genome = synth.make_genome(G, 20250, device=dev)
rs = synth.make_reads(genome, pairs, 20267)
del genome
torch.cuda.synchronize(); torch.cuda.empty_cache()
d = Dfk(K=48, device=0)
d.count_device(rs.packed, rs.base_off, rs.read_len, rs.pq_bytes, rs.pq_off, rs.bc)
st = d.stats(); print("solid", st["n_solid"], "ms", st["ms_total"], flush=True)
t0 = time.time(); g = d.graph_build(); t1 = time.time()
print("graph_build %.2f s: %s" % (t1 - t0, g), flush=True)
for rep in range(2):
    t0 = time.time(); p = d.paths_build_device(rs.packed, rs.base_off, rs.read_len, rs.pq_bytes, rs.pq_off); t1 = time.time()
    print("paths_build %.2f s (device %.3f s): %s" % (t1 - t0, d.stats()["us_paths"] * 1e-6, p), flush=True)
with tempfile.TemporaryDirectory(dir="/dev/shm") as t:
    t0 = time.time(); d.paths_index_write(t); t1 = time.time(); n = d.dups_write(t + "/a.dup"); t2 = time.time()
    print("paths_index %.2f s, mark_dups %.2f s (%d pairs)" % (t1 - t0, t2 - t1, n), flush=True)
