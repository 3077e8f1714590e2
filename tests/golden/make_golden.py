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


def make_hot(seed, n_reads):
    """Reads that all contain AGTACGGTATGCTCAC -- the one 16-mer whose minimizer rank is 0 in libdfk (dfk_device.h) --
    so that thousands of distinct k-mers share ONE fine bucket, plus a diverged repeat family and exact duplicates
    (so that solid k-mers exist): the input of the sub-pass and HBM-table paths."""
    rng = np.random.default_rng(seed)
    hot = np.array(["ACGT".index(ch) for ch in "AGTACGGTATGCTCAC"], np.uint8)
    elem = rng.integers(0, 4, 100, dtype=np.uint8)
    reads, quals = [], []
    for i in range(n_reads):
        if i % 5 == 4:                                   # a copy of the family element, 5 % diverged
            r = elem.copy(); hit = rng.random(100) < 0.05
            r[hit] = (r[hit] + rng.integers(1, 4, int(hit.sum()))) & 3
        else:
            r = rng.integers(0, 4, 100, dtype=np.uint8)
            r[42:58] = hot
        if i % 3 == 0 and reads:
            r = reads[-1].copy()
        if rng.random() < 0.3:
            r = (3 - r[::-1]).astype(np.uint8)
        reads.append(r); quals.append(np.full(100, 30, np.uint8))
    n_unbar = 2 * (n_reads // 16)
    bcs = np.sort(np.concatenate([np.zeros(n_unbar // 2, int), rng.integers(1, 9, (n_reads - n_unbar) // 2)]))
    bci = np.concatenate([[0], np.cumsum(np.bincount(bcs, minlength=9) * 2)]).astype(np.int64)
    return reads, quals, bci


GRAPH_FILES = ("a.k", "a.fastb", "a.hbv", "a.hbx", "a.kmers", "a.inv", "a.to_left", "a.to_right", "a.paths")


def refdrv(*args):
    """Run oracle/_ref/refdrv and READ what it prints: a MapReduceEngine run that overflowed a buffer says so
    ("There were N buffer overflows", MapReduceEngine.h:533-538) and has dropped barcodes -- not a golden run
    (caveat 2)."""
    out = subprocess.run([REFDRV, *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if out.returncode != 0 or "buffer overflow" in out.stdout:
        sys.stderr.write(out.stdout)
        raise SystemExit(f"refdrv {args[0]}: exit code {out.returncode}" + (", MapReduceEngine buffer overflows" if "buffer overflow" in out.stdout else ""))
    return out.stdout


def make_special(seed):
    """Error-free reads over three small sequences that exercise the corners of the edge builder: a circular
    chromosome (a cycle without branches: simpleCircle / canonicalizeCircle, BuildReadQGraph48.cc:338-392), a sequence
    with a reverse-complement-palindromic 48-mer in its middle (a one-k-mer edge that stops the walks on both sides),
    and two sequences sharing a 120-base stretch (branch vertices)."""
    rng = np.random.default_rng(seed)
    seqs = []
    circle = rng.integers(0, 4, 400, dtype=np.uint8)
    seqs.append(np.concatenate([circle, circle, circle[:100]]))            # reads wrap around twice
    half = rng.integers(0, 4, 24, dtype=np.uint8)
    pal = np.concatenate([half, (3 - half[::-1]).astype(np.uint8)])
    seqs.append(np.concatenate([rng.integers(0, 4, 150, dtype=np.uint8), pal, rng.integers(0, 4, 150, dtype=np.uint8)]))
    shared = rng.integers(0, 4, 120, dtype=np.uint8)
    for _ in range(2):
        seqs.append(np.concatenate([rng.integers(0, 4, 130, dtype=np.uint8), shared, rng.integers(0, 4, 130, dtype=np.uint8)]))
    reads, quals = [], []
    for s in seqs:
        for pos in range(0, len(s) - 100 + 1, 7):
            for rep in range(3):
                r = s[pos:pos + 100].copy()
                if (pos + rep) % 2:
                    r = (3 - r[::-1]).astype(np.uint8)
                reads.append(r); quals.append(np.full(100, 30, np.uint8))
    if len(reads) % 2:
        reads.append(reads[-1].copy()); quals.append(quals[-1].copy())
    bci = np.array([0, 0, len(reads)], np.int64)                           # one barcode; run without the barcode test
    return reads, quals, bci


def make_pathy(seed, G=9000, n_pairs=2600):
    """A diploid genome with SNP bubbles, a diverged repeat family and a tandem repeat; reads of 100-151 bases from both
    haplotypes and strands, 1 % errors that mostly carry low qualities, some exact duplicates (solid error k-mers: tips)
    and a few haplotype-switching reads: paths cross several short edges, end in gaps that the extension rules must
    resolve, and jump between repeat copies."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 4, G, dtype=np.uint8)
    fam = rng.integers(0, 4, 260, dtype=np.uint8)
    for s in (1200, 4100, 7000):                              # a diverged repeat family (3 % apart)
        c = fam.copy(); hit = rng.random(260) < 0.03
        c[hit] = (c[hit] + rng.integers(1, 4, int(hit.sum()))) & 3
        a[s:s + 260] = c
    unit = rng.integers(0, 4, 37, dtype=np.uint8)
    a[5200:5200 + 37 * 6] = np.tile(unit, 6)                  # a tandem repeat shorter than a read
    b = a.copy()
    pos = 300
    while pos < G - 300:                                      # heterozygous SNPs, some closer than K, some far apart
        b[pos] = (b[pos] + rng.integers(1, 4)) & 3
        pos += int(rng.choice([20, 35, 60, 90, 130, 200, 400]))
    # an exact repeat of K-1 = 47 bases at two loci with different flanks: no k-mer is shared, so the graph does not branch
    # there -- but a read that enters at one locus and leaves at the other (its switch inside a Q2 tail, where createDict
    # does not look) steps from the middle of one edge into the middle of another: the adjacency check of algorithmTwo
    l1, l2 = 2500, 6100
    for h in (a, b):
        h[l2:l2 + 47] = h[l1:l1 + 47]
        for d in (-1, 47):
            if h[l1 + d] == h[l2 + d]: h[l2 + d] = (h[l2 + d] + 1) & 3
    haps = (a, b)
    reads, quals = [], []
    for p in range(n_pairs):
        for mate in range(2):
            L = int(rng.choice([100, 100, 100, 120, 150, 151]))
            start = int(rng.integers(0, G - L))
            h = haps[int(rng.integers(0, 2))]
            r = h[start:start + L].copy()
            if rng.random() < 0.04:                           # switches haplotype in the middle
                cut = int(rng.integers(20, L - 20)); r[cut:] = haps[1 - (h is b)][start + cut:start + L]
            q = rng.choice([37, 37, 37, 35, 30, 25], L).astype(np.uint8)
            err = rng.random(L) < 0.01
            if rng.random() < 0.25:
                err[int(rng.integers(0, L))] = True           # at least one error in a quarter of the reads
            r[err] = (r[err] + rng.integers(1, 4, int(err.sum()))) & 3
            lowq = err & (rng.random(L) < 0.7)
            q[lowq] = rng.choice([2, 2, 8, 12, 15], int(lowq.sum()))
            if rng.random() < 0.1:
                t = int(rng.integers(1, 12)); q[L - t:] = 2
            if rng.random() < 0.5:
                r = (3 - r[::-1]).astype(np.uint8); q = q[::-1].copy()
            if rng.random() < 0.06 and reads:                 # an exact duplicate of an earlier read (errors and all)
                k = int(rng.integers(0, len(reads))); r = reads[k].copy(); q = quals[k].copy()
            reads.append(r); quals.append(q)
    for i in range(8):                                        # the switching reads (they replace ordinary ones)
        q = np.full(100, 37, np.uint8); q[80:] = 2            # the switch lies in the 3' tail that the trim removes
        if i % 2 == 0: r = np.concatenate([a[l1 - 38 - i:l1 + 47], a[l2 + 47:l2 + 47 + 15 - i]])
        else: r = (3 - np.concatenate([a[l2 - 15 + i:l2], a[l1:l1 + 85 + i]])[::-1]).astype(np.uint8)
        assert len(r) == 100
        reads[40 + 2 * i] = r; quals[40 + 2 * i] = q
    n_unbar = n_pairs // 10
    bcs = np.sort(np.concatenate([np.zeros(n_unbar, int), rng.integers(1, 25, n_pairs - n_unbar)]))
    bci = np.concatenate([[0], np.cumsum(np.bincount(bcs, minlength=25) * 2)]).astype(np.int64)
    return reads, quals, bci


def make_pathy2(seed, n_blocks=9, G1=9000, pairs_per_block=1300):
    """The pather's second input: n_blocks independent genomes of make_pathy's kind (each with its own SNP bubbles, repeat family,
    tandem repeat and K-1 repeat with its eight switching reads), and in every block a ninth of the reads with a tail of 25-45
    random bases of HIGH quality -- the path ends before the tail, the overhang is long and no edge at the end vertex explains
    it.  Made so that every one of the 16 rules of algorithmTwo / the extensions fires at least 50 times (pathy2_rules.txt)."""
    rng = np.random.default_rng(seed)
    reads, quals = [], []
    for b in range(n_blocks):
        r, q, _ = make_pathy(seed * 100 + b, G1, pairs_per_block)
        for i in range(60, len(r), 9):
            L = len(r[i]); t = int(rng.integers(25, 46))
            r[i] = r[i].copy(); r[i][L - t:] = rng.integers(0, 4, t); q[i] = q[i].copy(); q[i][L - t:] = 37
        reads += r; quals += q
    n_pairs = len(reads) // 2
    n_unbar = n_pairs // 10
    bcs = np.sort(np.concatenate([np.zeros(n_unbar, int), rng.integers(1, 60, n_pairs - n_unbar)]))
    bci = np.concatenate([[0], np.cumsum(np.bincount(bcs, minlength=60) * 2)]).astype(np.int64)
    return reads, quals, bci


def make_frag(seed, G=26000, n_pairs=3000):
    """A fragmented graph (a SNP every ~70 bases of a diploid genome: more than 870 HBV edges, below which the reference's
    writePathsIndex overruns) read by PAIRS of 100 bases with inserts of 250-400, a tenth of the pairs PCR duplicates of an
    earlier pair: same bases, qualities drawn again (so the pair with the higher quality sum stays) or kept (a tie: the
    earlier pair stays)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 4, G, dtype=np.uint8)
    b = a.copy()
    pos = 200
    while pos < G - 200:
        b[pos] = (b[pos] + rng.integers(1, 4)) & 3
        pos += int(rng.choice([30, 50, 70, 90, 140]))
    haps = (a, b)
    reads, quals = [], []
    for p in range(n_pairs):
        if p > 10 and rng.random() < 0.10:                               # a duplicate of an earlier pair
            k = int(rng.integers(0, p))
            r1, r2 = reads[2 * k].copy(), reads[2 * k + 1].copy()
            if rng.random() < 0.3: q1, q2 = quals[2 * k].copy(), quals[2 * k + 1].copy()
            else: q1, q2 = (np.full(100, rng.choice([37, 35, 30, 25]), np.uint8) for _ in range(2))
        else:
            ins = int(rng.integers(250, 400)); start = int(rng.integers(0, G - ins)); h = haps[int(rng.integers(0, 2))]
            r1 = h[start:start + 100].copy(); r2 = (3 - h[start + ins - 100:start + ins][::-1]).astype(np.uint8)
            if rng.random() < 0.5: r1, r2 = r2, r1
            q1, q2 = (np.full(100, rng.choice([37, 37, 35, 30]), np.uint8) for _ in range(2))
            for r, q in ((r1, q1), (r2, q2)):
                err = rng.random(100) < 0.004
                r[err] = (r[err] + rng.integers(1, 4, int(err.sum()))) & 3; q[err] = 12
        reads += [r1, r2]; quals += [q1, q2]
    n_unbar = n_pairs // 10
    bcs = np.sort(np.concatenate([np.zeros(n_unbar, int), rng.integers(1, 31, n_pairs - n_unbar)]))
    bci = np.concatenate([[0], np.cumsum(np.bincount(bcs, minlength=31) * 2)]).astype(np.int64)
    return reads, quals, bci


def run_graph(head, out, K, use_bc, min_bc, min_freq, dest, extra=()):
    """refdrv graph: dict, then the unipath edges through the real KmerDict, a.<K>/ through the real digraphE, and a.paths:
    every read pathed through the real KmerDict / digraphE and written by the real ReadPathVec writer (row f-2)."""
    os.makedirs(out, exist_ok=True)
    log = refdrv("graph", K, head, out, 7, min_freq, min_bc, use_bc, 4, 0)
    print("   ", " | ".join(l.strip() for l in log.splitlines() if l.startswith(("paths:", "dups:")) and not l.startswith("paths:  ")))
    os.makedirs(dest, exist_ok=True)
    for f in GRAPH_FILES + tuple(extra):
        open(os.path.join(dest, f), "wb").write(open(os.path.join(out, f"a.{K}", f), "rb").read())
    post = np.fromfile(out + "/solid.bin", ENTRY)
    n_edges = int(np.frombuffer(open(os.path.join(dest, "a.kmers"), "rb").read()[8:16], "<u8")[0])
    subprocess.check_call(["rm", "-rf", out])
    return post, n_edges


def run_dict(head, out, K, use_bc, min_bc, min_freq=3, ign=0):
    os.makedirs(out, exist_ok=True)
    refdrv("dict", K, head, out, 7, min_freq, min_bc, use_bc, 4, ign)
    post = np.fromfile(out + "/solid.bin", ENTRY)
    kv = open(out + "/kmers.kvec", "rb").read()
    assert kv[:8] == b"BINWRITE"
    n = struct.unpack("<Q", kv[8:16])[0]
    pre = np.frombuffer(kv, ENTRY, count=n, offset=16).copy()
    pre["bc"] = -1; pre["pad"] = 0                       # tempBC / pad are arbitrary in the reference
    pre = pre[np.lexsort((pre["w1"], pre["w0"]))]
    good = np.fromfile(out + "/goodlens.u32", np.uint32)
    spec = np.loadtxt(out + "/spectrum.txt", dtype=np.int64, ndmin=1)
    subprocess.check_call(["rm", "-rf", out])
    return good, pre, post, spec


# filter variants of createDict (BuildReadQGraph48.cc:104-132,167-174,313) on the golden reads: (tag, min_freq, min_bc, use_bc, ign_bc_below)
VARIANTS = [("minbc0", 3, 0, 1, 0), ("minbc3", 3, 3, 1, 0), ("minbc4", 3, 4, 1, 0), ("minbc5", 3, 5, 1, 0), ("minbc6_minfreq2", 2, 6, 1, 0),
            ("minbc8", 3, 8, 1, 0), ("minfreq1", 1, 2, 1, 0),
            ("minfreq2", 2, 2, 1, 0), ("minfreq5", 5, 2, 1, 0), ("ign600", 3, 2, 1, 600), ("ign600_minbc3_minfreq2", 2, 3, 1, 600),
            ("ign_all", 5, 2, 1, 10 ** 9), ("minfreq1_nobc", 1, 0, 0, 0)]


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
        good, pre, post, spec = run_dict(head, os.path.join(HERE, "tmp_" + tag), K, use_bc, min_bc)
        np.savez_compressed(os.path.join(HERE, f"expect_{tag}.npz"), good_len=good, solid_post=post, solid_pre=pre, spectrum=spec)
        print(tag, "solid", len(post), "ctx rewritten by adjacency", int((post["count_ctx"] != pre["count_ctx"]).sum()))
    # The filter variants: what is kept is small -- the solid count, the spectrum and the order-independent digest of
    # the entries before and after recomputeAdjacencies (superplus_amd.dfk.digest_of; every parity test ties that
    # digest to the entries themselves).
    import json
    from superplus_amd.dfk import digest_of
    var = {}
    for tag, min_freq, min_bc, use_bc, ign in VARIANTS:
        good, pre, post, spec = run_dict(head, os.path.join(HERE, "tmp_" + tag), 48, use_bc, min_bc, min_freq, ign)
        var[tag] = dict(K=48, min_freq=min_freq, min_bc=min_bc, use_bc=use_bc, ign_bc_below=ign, n_solid=len(post),
                        digest_pre=[str(x) for x in digest_of(pre)], digest_post=[str(x) for x in digest_of(post)],
                        spectrum=[int(x) for x in spec])
        print(tag, "solid", len(post))
    json.dump(var, open(os.path.join(HERE, "variants.json"), "w"), indent=1)
    # one hot-minimizer input, full entries
    reads, quals, bci = make_hot(777, 1500)
    raw = os.path.join(HERE, "hot.raw")
    write_raw(raw, reads, quals)
    hot = os.path.join(HERE, "hot")
    subprocess.check_call([REFDRV, "mkreads", raw, hot], stdout=subprocess.DEVNULL)
    os.remove(raw)
    feudal.write_bci(hot + ".bci", bci)
    good, pre, post, spec = run_dict(hot, os.path.join(HERE, "tmp_hot"), 48, 1, 2, 2)
    np.savez_compressed(os.path.join(HERE, "expect_hot_k48_minfreq2.npz"), good_len=good, solid_post=post, solid_pre=pre, spectrum=spec)
    print("hot solid", len(post))
    # The graph half (row f-1): a.<K>/ files for the golden reads at K = 48/40/60, the hot-minimizer input, and a
    # special input with a cycle, a palindromic k-mer and branches.
    for tag, K, use_bc, min_bc in (("k48", 48, 1, 2), ("k40_nobc", 40, 0, 0), ("k60_nobc", 60, 0, 0)):
        post, ne = run_graph(head, os.path.join(HERE, "tmp_g" + tag), K, use_bc, min_bc, 3, os.path.join(HERE, "graph_" + tag))
        print("graph", tag, "solid", len(post), "HBV edges", ne)
    post, ne = run_graph(hot, os.path.join(HERE, "tmp_ghot"), 48, 1, 2, 2, os.path.join(HERE, "graph_hot_k48_minfreq2"))
    print("graph hot: solid", len(post), "HBV edges", ne)
    # ParseBarcodedFastqs (row f-3): a small fastq.gz pair through the reference's OWN binary (oracle/_ref/ParseBarcodedFastqs,
    # built from 10X/ParseBarcodedFastqs.cc where it lies; single-threaded, see oracle/build_ref.sh)
    from tests.fastq_synth import make_fastq
    pbf = os.path.join(HERE, "pbf")
    os.makedirs(pbf, exist_ok=True)
    make_fastq(pbf + "/r_1.fq.gz", pbf + "/r_2.fq.gz", 300, 7, n_bc=12, ragged=True)
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "ParseBarcodedFastqs"), "FASTQS={" + pbf + "/r_1.fq.gz," + pbf + "/r_2.fq.gz}",
                           "OUT_HEAD=" + pbf + "/reads", "NUM_THREADS=1", "NUM_BUCKETS=4"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads, quals, bci = make_special(99)
    raw = os.path.join(HERE, "special.raw")
    write_raw(raw, reads, quals)
    sp = os.path.join(HERE, "special")
    subprocess.check_call([REFDRV, "mkreads", raw, sp], stdout=subprocess.DEVNULL)
    os.remove(raw)
    feudal.write_bci(sp + ".bci", bci)
    post, ne = run_graph(sp, os.path.join(HERE, "tmp_gsp"), 48, 0, 0, 3, os.path.join(HERE, "graph_special_k48"))
    np.savez_compressed(os.path.join(HERE, "expect_special_k48_nobc.npz"), solid_post=post)
    print("graph special: solid", len(post), "HBV edges", ne)
    # Row f-2: an input made for the read pather -- bubbles, tips, repeat copies, a K-1 repeat -- on which every rule of
    # HBVPather::algorithmTwo and of the two extensions fires (refdrv prints how often; see oracle/ref_graph.cc)
    reads, quals, bci = make_pathy(5, 9000, 1800)
    raw = os.path.join(HERE, "pathy.raw")
    write_raw(raw, reads, quals)
    pa = os.path.join(HERE, "pathy")
    subprocess.check_call([REFDRV, "mkreads", raw, pa], stdout=subprocess.DEVNULL)
    os.remove(raw)
    feudal.write_bci(pa + ".bci", bci)
    os.makedirs(os.path.join(HERE, "tmp_rules"), exist_ok=True)
    open(os.path.join(HERE, "pathy_rules.txt"), "w").write(
        "".join(l + "\n" for l in refdrv("graph", 48, pa, os.path.join(HERE, "tmp_rules"), 7, 3, 2, 1, 4, 0).splitlines() if l.startswith(("paths:", "graph:"))))
    subprocess.check_call(["rm", "-rf", os.path.join(HERE, "tmp_rules")])
    post, ne = run_graph(pa, os.path.join(HERE, "tmp_gpa"), 48, 1, 2, 3, os.path.join(HERE, "graph_pathy_k48"))
    np.savez_compressed(os.path.join(HERE, "expect_pathy_k48.npz"), solid_post=post)
    print("graph pathy: solid", len(post), "HBV edges", ne)
    # ... and a second, larger one on which every rule fires at least 50 times (the round-3 review: three rules fired < 10 times on
    # pathy); with the files of row f-4 as well
    reads, quals, bci = make_pathy2(7)
    raw = os.path.join(HERE, "pathy2.raw")
    write_raw(raw, reads, quals)
    pa2 = os.path.join(HERE, "pathy2")
    subprocess.check_call([REFDRV, "mkreads", raw, pa2], stdout=subprocess.DEVNULL)
    os.remove(raw)
    feudal.write_bci(pa2 + ".bci", bci)
    os.makedirs(os.path.join(HERE, "tmp_rules"), exist_ok=True)
    rules = [l for l in refdrv("graph", 48, pa2, os.path.join(HERE, "tmp_rules"), 7, 3, 2, 1, 4, 0).splitlines() if l.startswith(("paths:", "graph:"))]
    open(os.path.join(HERE, "pathy2_rules.txt"), "w").write("".join(l + "\n" for l in rules))
    assert all(int(l.rsplit(None, 1)[1]) >= 50 for l in rules if l.startswith("paths:   ")), rules
    subprocess.check_call(["rm", "-rf", os.path.join(HERE, "tmp_rules")])
    post, ne = run_graph(pa2, os.path.join(HERE, "tmp_gpa2"), 48, 1, 2, 3, os.path.join(HERE, "graph_pathy2_k48"), extra=("a.paths.inv", "a.countsb", "a.dup"))
    np.savez_compressed(os.path.join(HERE, "expect_pathy2_k48.npz"), solid_post=post)
    print("graph pathy2: solid", len(post), "HBV edges", ne)
    # Row f-4: a fragmented graph (more than 870 edges: below that the reference's writePathsIndex overruns, SURVEY 8c caveat 1)
    # read by pairs with PCR duplicates: a.paths.inv, a.countsb, a.dup
    reads, quals, bci = make_frag(9, 26000, 3000)
    raw = os.path.join(HERE, "frag.raw")
    write_raw(raw, reads, quals)
    fr = os.path.join(HERE, "frag")
    subprocess.check_call([REFDRV, "mkreads", raw, fr], stdout=subprocess.DEVNULL)
    os.remove(raw)
    feudal.write_bci(fr + ".bci", bci)
    post, ne = run_graph(fr, os.path.join(HERE, "tmp_gfr"), 48, 1, 2, 3, os.path.join(HERE, "graph_frag_k48"), extra=("a.paths.inv", "a.countsb", "a.dup"))
    np.savez_compressed(os.path.join(HERE, "expect_frag_k48.npz"), solid_post=post)
    assert ne >= 870
    print("graph frag: solid", len(post), "HBV edges", ne)


if __name__ == "__main__":
    main()
