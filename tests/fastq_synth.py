"""Seeded synthetic stLFR fastq.gz pairs for the ParseBarcodedFastqs tests (names "@id#b1_b2_b3/1<TAB>...", barcode
0_0_0 = unbarcoded; split_barcode_PEXXX_42_unsort_reads.pl's format).  Harness only."""
import gzip

import numpy as np


def make_fastq(path1, path2, n_pairs, seed, n_bc=40, L=100, ragged=False):
    rng = np.random.default_rng(seed)
    G = rng.integers(0, 4, 5000)
    prev = None
    with gzip.open(path1, "wt") as f1, gzip.open(path2, "wt") as f2:
        for i in range(n_pairs):
            if rng.random() < 0.12:
                b = (0, 0, 0)
            else:
                k = int(rng.integers(1, n_bc + 1)); b = (k % 7 + 1, (k * 13) % 1500 + 1, (k * 101) % 1536 + 1)
            cur = {}
            for f, mate in ((f1, 1), (f2, 2)):
                l = int(rng.choice([L, L, L, 75, 151, 49])) if ragged else L
                pos = int(rng.integers(0, len(G) - l))
                s = "".join("ACGT"[x] for x in G[pos:pos + l])
                if rng.random() < 0.1:
                    j = int(rng.integers(0, l)); s = s[:j] + "N" + s[j + 1:]
                if rng.random() < 0.3 and prev is not None:
                    s = prev[mate]                                        # duplicates: ties in the per-barcode sort
                q = rng.choice([37, 37, 37, 30, 25, 12, 2], len(s))
                if rng.random() < 0.3:
                    q[:] = 35
                if rng.random() < 0.3:
                    q[-int(rng.integers(1, 20)):] = 2
                f.write(f"@r{i}#{b[0]}_{b[1]}_{b[2]}/{mate}\t{i}\t1\n{s}\n+\n{''.join(chr(33 + int(x)) for x in q)}\n")
                cur[mate] = s
            prev = cur
