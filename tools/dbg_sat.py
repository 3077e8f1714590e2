import sys, numpy as np
sys.path.insert(0, '.')
from superplus_amd import feudal
from superplus_amd.dfk import Dfk
n = 330_000
rng = np.random.default_rng(17)
extra = rng.integers(0, 4, (3000, 100), dtype=np.uint8)
extra[1000:2000] = extra[:1000]; extra[2000:] = extra[:1000]
packed = np.concatenate([np.zeros(25 * n, np.uint8), feudal.pack_bases(extra).reshape(-1)])
N = n + len(extra)
blk = np.array([100, (35 << 3) & 0xFF, 35 >> 5, 0], np.uint8)
rs = dict(packed=packed, base_off=(np.arange(N + 1, dtype=np.uint64) * 25), read_len=np.full(N, 100, np.uint32),
          pq_bytes=np.tile(blk, N), pq_off=(np.arange(N + 1, dtype=np.uint64) * 4), bc=(1 + np.arange(N) % 7).astype(np.int32))
for it in range(6):
    d = Dfk(K=48, keep_pre_adjacency=True)
    d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    st = d.stats(); s = d.solid()
    key = s["w0"].astype(object) * (1 << 64) + s["w1"].astype(object)
    dup = len(key) - len(set(key))
    top = s[s["w0"] == 0]
    print(it, "n_distinct", st["n_distinct"], "n_solid", st["n_solid"], "dups in solid", dup, "poly-A entries", [(hex(int(x["w1"])), hex(int(x["count_ctx"]))) for x in top], "overflow items", st["n_overflow_items"], flush=True)
    d.close()
