"""tools/df_determinism.py [PAIRS] [GENOME_MB] -- the DF stage twice on the same input files: every file it writes must come out
the same both times (the device's atomics order nothing that reaches a file: edges are renumbered canonically on the host,
paths are written in read order, the dictionary file in its own sorted order when asked)."""
import hashlib, os, shutil, subprocess, sys
import torch
sys.path.insert(0, '.')
import bench
from superplus_amd import synth
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000
G = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else int(pairs * 200 / 58)
dev = torch.device("cuda:0")
genome = synth.make_genome(G, 777, device=dev)
rs = synth.make_reads(genome, pairs, 778)
del genome
root = "/dev/shm/dfdet"
shutil.rmtree(root, ignore_errors=True); os.makedirs(root + "/in")
bench.write_read_files(rs, root + "/in/reads")
del rs
torch.cuda.synchronize(); torch.cuda.empty_cache()


def digest(d):
    out = {}
    for base, _, files in os.walk(d):
        for f in files:
            p = os.path.join(base, f)
            h = hashlib.md5()
            with open(p, "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""): h.update(blk)
            out[os.path.relpath(p, d)] = (os.path.getsize(p), h.hexdigest())
    return out


runs = []
for k in range(2):
    r = f"{root}/run{k}"
    os.makedirs(r)
    for e in (".fastb", ".qualp", ".bci"): os.link(root + "/in/reads" + e, r + "/reads" + e)
    p = subprocess.run([os.path.join("superplus_amd", "DF"), f"ROOT={r}", f"LR={r}/reads.fastb", "PIPELINE=cs", "ALIGN=False", "NUM_THREADS=16",
                        "MAX_MEM_GB=640", "KVEC=True", "KVEC_SORTED=True"], capture_output=True, text=True)
    if p.returncode: raise SystemExit(p.stdout[-2000:] + p.stderr[-2000:])
    runs.append(digest(r + "/GapToy"))
    shutil.rmtree(r + "/GapToy")
a, b = runs
diff = sorted(f for f in set(a) | set(b) if a.get(f) != b.get(f) and not f.endswith('the_command'))   # (the_command holds the run's own ROOT)
print(f"{len(a)} files, {sum(s for s, _ in a.values()) / 1e9:.1f} GB; differing: {diff if diff else 'none'}")
shutil.rmtree(root, ignore_errors=True)
sys.exit(1 if diff else 0)
