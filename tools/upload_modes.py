"""tools/upload_modes.py [PAIRS] -- dfk_count's upload from mapped read files (the DF fast path) three ways: no hint, the hint
with a descriptor (the library reads the file), the hint without (it reads the memory and drops the pages).  Prints the
library's own upload time for each and how long unmapping took afterwards."""
import mmap, os, struct, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from superplus_amd import synth
from superplus_amd.dfk import Dfk
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda:0")
genome = synth.make_genome(int(pairs * 200 / 58), 20250, device=dev)
rs = synth.make_reads(genome, pairs, 20267)
del genome
head = "/dev/shm/upl_reads"
bench.write_read_files(rs, head)
n = rs.n_reads
bc = rs.bc.cpu().numpy()
del rs
torch.cuda.synchronize(); torch.cuda.empty_cache()


def views(path, fixed):
    fd = os.open(path, os.O_RDONLY)
    size = os.fstat(fd).st_size
    m = mmap.mmap(fd, size, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
    if os.environ.get("UPL_SEQ"): m.madvise(mmap.MADV_SEQUENTIAL)
    whole = np.frombuffer(m, np.uint8)
    cnt, _, _, _, _, var_tab, fixed_off = struct.unpack("<IBBBBQQ", bytes(whole[:24]))
    off = whole[var_tab:var_tab + 8 * (n + 1)].view(np.uint64)
    fx = whole[fixed_off:fixed_off + 4 * n].view(np.uint32) if fixed else None
    return fd, m, whole, off, fx


for mode in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("plain", "file", "drop", "plain")):
    fb = views(head + ".fastb", True); qp = views(head + ".qualp", False)
    d = Dfk(K=48)
    if mode != "plain":
        d.hint_file_range(fb[2], fb[0] if mode == "file" else -1)
        d.hint_file_range(qp[2], qp[0] if mode == "file" else -1)
    t0 = time.time()
    d.count(fb[2], fb[3], fb[4], qp[2], qp[3], bc)
    st = d.stats()
    t1 = time.time()
    d.close()
    fd1, m1, w1, o1, f1 = fb; fd2, m2, w2, o2, f2 = qp
    del fb, qp, w1, o1, f1, w2, o2, f2
    t2 = time.time(); m1.close(); m2.close(); t3 = time.time()
    os.close(fd1); os.close(fd2)
    print(f"{mode:5s}: upload {st['ms_upload'] / 1e3:.2f} s of count {t1 - t0:.2f} s, solid {st['n_solid']}, unmapping {t3 - t2:.2f} s", flush=True)
for e in (".fastb", ".qualp", ".bci"):
    try: os.remove(head + e)
    except OSError: pass
