"""Same reads, different pass geometries / HBM budgets: n_distinct, n_solid and the dictionary digest must not move.
usage: python tools/dbg_determinism.py GENOME_MB COPIES COVERAGE [oracle] [passes,budget_gb ...]
(env switches select the hot-bucket path; `oracle` runs oracle/_ref/refdrv on the same reads and diffs the dictionaries)"""
import os, subprocess, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, '.')
from superplus_amd import synth, feudal
from superplus_amd.dfk import Dfk, ENTRY_DTYPE, digest_of
G = int(float(sys.argv[1]) * 1e6); copies = int(sys.argv[2]); cov = float(sys.argv[3])
rest = sys.argv[4:]
oracle = bool(rest and rest[0] == "oracle")
if oracle: rest = rest[1:]
cfgs = [tuple(float(x) for x in a.split(",")) for a in rest] or [(0, 0)]
dev = torch.device("cuda:0")
genome = synth.make_genome(G, 20250, device=dev, family_copies=copies, low_complexity_frac=0.01)
rs = synth.make_reads(genome, int(cov * G / 200), 20267, ragged_frac=0.25)
del genome
torch.cuda.synchronize(); torch.cuda.empty_cache()
shard = (rs.packed, rs.base_off, rs.read_len, rs.pq_bytes, rs.pq_off, rs.bc)
truth = None
CACHE = f"/dev/shm/dbg_truth_{sys.argv[1]}_{copies}_{sys.argv[3]}.npy"
if oracle and os.path.exists(CACHE):
    truth = np.load(CACHE)
    tk = None
elif oracle:
    t0 = time.time()
    with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
        feudal.write_fastb(d + "/s.fastb", rs.packed.cpu().numpy(), rs.base_off.cpu().numpy().astype(np.uint64), rs.read_len.cpu().numpy().astype(np.uint32))
        feudal.write_qualp(d + "/s.qualp", rs.pq_bytes.cpu().numpy(), rs.pq_off.cpu().numpy().astype(np.uint64))
        b64 = rs.bc.cpu().numpy().astype(np.int64)
        bci = np.concatenate([[0], np.cumsum(np.bincount(b64, minlength=int(b64.max()) + 1))]).astype(np.int64)
        feudal.write_bci(d + "/s.bci", bci)
        os.makedirs(d + "/o")
        pr = subprocess.Popen(["oracle/_ref/refdrv", "dict", "48", d + "/s", d + "/o", "7", "3", "2", "1", str(min(32, os.cpu_count()))], stdout=subprocess.DEVNULL)
        while pr.poll() is None:
            time.sleep(30); print(f"  (oracle running, {time.time() - t0:.0f} s)", flush=True)
        assert pr.returncode == 0, f"refdrv exited {pr.returncode}"
        truth = np.fromfile(d + "/o/solid.bin", ENTRY_DTYPE)
    np.save(CACHE, truth)
    print(f"oracle: {len(truth)} solid, digest {digest_of(truth)}, {time.time() - t0:.0f} s", flush=True)
    tk = np.stack([truth["w0"], truth["w1"]], 1)
def keyset(a):
    v = np.ascontiguousarray(np.stack([a["w0"], a["w1"]], 1)).view([("a", "<u8"), ("b", "<u8")]).reshape(-1)
    return v
for passes, gb in cfgs:
    d = Dfk(K=48, device=0, passes=int(passes), hbm_budget_bytes=int(gb * 1e9))
    for rep in range(2):
        d.count_device(*shard)
        st = d.stats()
        print(f"passes={int(passes)} budget={gb} rep={rep}: n_passes {st['n_passes']} n_inst {st['n_inst']} n_distinct {st['n_distinct']} "
              f"n_solid {st['n_solid']} overflow_items {st['n_overflow_items']} digest {d.digest()} ms {st['ms_count']:.0f} fb {st['ms_fallback']:.0f}", flush=True)
        if truth is not None and d.digest() != digest_of(truth):
            s = d.solid()
            a, b = keyset(s), keyset(truth)
            ua, ca = np.unique(a, return_counts=True)
            print(f"   MISMATCH: ours {len(s)} entries ({int((ca > 1).sum())} keys more than once), truth {len(truth)}; "
                  f"extra keys {len(np.setdiff1d(ua, b))}, missing keys {len(np.setdiff1d(b, ua))}", flush=True)
            ex, mi = np.setdiff1d(ua, b), np.setdiff1d(b, ua)
            for k in ex[:12]: print("      extra  ", [hex(int(x)) for x in s[a == k][0].tolist()], flush=True)
            for k in mi[:12]: print("      missing", [hex(int(x)) for x in truth[b == k][0].tolist()], flush=True)
            import collections
            print("      extra counts:", collections.Counter((s[np.isin(a, ex)]["count_ctx"] >> 8).tolist()).most_common(8))
            print("      missing counts:", collections.Counter((truth[np.isin(b, mi)]["count_ctx"] >> 8).tolist()).most_common(8))
            so, to = s[np.argsort(a, kind='stable')], truth[np.argsort(b, kind='stable')]
            if len(so) == len(to):
                diff = np.nonzero((so["count_ctx"] != to["count_ctx"]))[0]
                print(f"   same keys: {bool((keyset(so) == keyset(to)).all())}; entries whose count/context word differs: {len(diff)}", flush=True)
                for i in diff[:8]: print("     ", [hex(int(x)) for x in so[i].tolist()], [hex(int(x)) for x in to[i].tolist()])
            dupk = ua[ca > 1][:4]
            for k in dupk: print("      dup", [[hex(int(x)) for x in e.tolist()] for e in s[a == k]])
    d.close()
