#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerate the golden vectors (build container only).

Inputs are made here (seeded numpy), then ENCODED AND WRITTEN BY THE REFERENCE'S OWN CLASSES
(BaseVec, PQVecEncoder, feudal writer) through oracle/_ref/refdrv, and the expected outputs
come from refdrv's `dict` run: the reference's KMer/KMerContext/MapReduceEngine/KmerVec/
BinaryWriter/KmerDict::recomputeAdjacencies compiled in place from /root/reference (glue
restated; see oracle/ref_driver.cc).  Only data is committed: no reference source or script.

  python tests/golden/make_golden.py        (needs oracle/_ref/refdrv: oracle/build_ref.sh)
"""
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REFDRV = os.path.join(ROOT, "oracle", "_ref", "refdrv")
ENTRY = np.dtype([("w0", "<u8"), ("w1", "<u8"), ("edge_id", "<u4"), ("count_ctx", "<u4"), ("bc", "<i4"), ("pad", "<u4")])


def make_raw(seed, G, n_pairs):
    """Ragged, messy reads: lengths 30..151, per-base qualities with low-quality stretches."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, G, dtype=np.uint8)
    genome[500:900] = genome[100:500]                       # a repeat
    genome[2000:2060] = 0                                   # poly-A
    reads, quals = [], []
    n_unbar = n_pairs // 8
    n_bc = 12
    bcs = np.sort(np.concatenate([np.zeros(n_unbar, int), rng.integers(1, n_bc + 1, n_pairs - n_unbar)]))
    for p in range(n_pairs):
        for mate in range(2):
            L = int(rng.choice([100, 100, 100, 150, 151, 75, 60, 49, 48, 30]))
            pos = int(rng.integers(0, G - L))
            r = genome[pos:pos + L].copy()
            if rng.random() < 0.5:
                r = (3 - r[::-1]).astype(np.uint8)
            err = rng.random(L) < 0.01
            r[err] = (r[err] + rng.integers(1, 4, int(err.sum()))) & 3
            q = rng.choice([37, 37, 35, 30, 25, 12, 8, 7], L).astype(np.uint8)
            if rng.random() < 0.4:
                t = int(rng.integers(1, 25)); q[L - t:] = rng.integers(0, 7, t)
            if rng.random() < 0.15:
                a = int(rng.integers(0, L)); q[a:a + 3] = 2
            if rng.random() < 0.05:
                q[:] = 40                                   # constant-quality read: single nBits=0 block
            if rng.random() < 0.03:
                q[:] = 3                                    # all bad
            reads.append(r); quals.append(q)
    bci = np.concatenate([[0], np.cumsum(np.bincount(bcs, minlength=n_bc + 1) * 2)]).astype(np.int64)
    return reads, quals, bci


def write_raw(path, reads, quals):
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(reads)))
        for r, q in zip(reads, quals):
            f.write(struct.pack("<I", len(r))); f.write(r.tobytes()); f.write(q.tobytes())


def main():
    from superplus_amd import feudal
    if not os.path.exists(REFDRV):
        raise SystemExit("oracle/_ref/refdrv missing: run oracle/build_ref.sh in the build container")
    with open(os.path.join(HERE, "kat.txt"), "w") as f:
        f.write(subprocess.check_output([REFDRV, "kat"], text=True))
    reads, quals, bci = make_raw(4242, 9000, 900)
    raw = os.path.join(HERE, "reads.raw")
    write_raw(raw, reads, quals)
    head = os.path.join(HERE, "reads")
    subprocess.check_call([REFDRV, "mkreads", raw, head], stdout=subprocess.DEVNULL)
    feudal.write_bci(head + ".bci", bci)
    side = os.path.join(HERE, "side")
    os.makedirs(side, exist_ok=True)
    subprocess.check_call([REFDRV, "side", head, side])      # .lens/.qhist/.dti/subsam.*/.1000.* via the reference's BinaryWriter, feudal writers and randomx()
    side07 = os.path.join(HERE, "side_frac07")
    os.makedirs(side07, exist_ok=True)
    subprocess.check_call([REFDRV, "side", head, side07, "0.7"])   # the same with LR_SELECT_FRAC = 0.7 (+ the selected reads themselves)
    for K, use_bc, min_bc, tag in ((48, 1, 2, "k48"), (48, 1, 1, "k48_minbc1"), (40, 0, 0, "k40_nobc"), (60, 0, 0, "k60_nobc")):
        out = os.path.join(HERE, "tmp_" + tag)
        os.makedirs(out, exist_ok=True)
        subprocess.check_call([REFDRV, "dict", str(K), head, out, "7", "3", str(min_bc), str(use_bc), "4"],
                              stdout=subprocess.DEVNULL)
        post = np.fromfile(out + "/solid.bin", ENTRY)
        kv = open(out + "/kmers.kvec", "rb").read()
        assert kv[:8] == b"BINWRITE"
        n = struct.unpack("<Q", kv[8:16])[0]
        pre = np.frombuffer(kv, ENTRY, count=n, offset=16).copy()
        pre["bc"] = -1; pre["pad"] = 0                       # tempBC / pad are arbitrary in the reference
        pre = pre[np.lexsort((pre["w1"], pre["w0"]))]
        np.savez_compressed(os.path.join(HERE, f"expect_{tag}.npz"),
                            good_len=np.fromfile(out + "/goodlens.u32", np.uint32),
                            solid_post=post, solid_pre=pre,
                            spectrum=np.loadtxt(out + "/spectrum.txt", dtype=np.int64, ndmin=1))
        print(tag, "solid", len(post), "ctx rewritten by adjacency", int((post["count_ctx"] != pre["count_ctx"]).sum()))
        subprocess.check_call(["rm", "-rf", out])


if __name__ == "__main__":
    main()
