"""Shared helpers for the dfk tests."""
import numpy as np
import torch

from superplus_amd import synth


def make_set(seed, genome_size, n_pairs, read_len=100, **kw):
    genome = synth.make_genome(genome_size, seed, repeat_frac=kw.pop("repeat_frac", 0.02))
    return synth.make_reads(genome, n_pairs, seed + 1, read_len=read_len, **kw).numpy()


def make_long_set(seed, genome_size, n_pairs, read_len=300, err=0.005):
    """Pairs of reads longer than one PQVec block (nQs is a byte: 255), built on the host with the small-test encoder
    (feudal.pq_encode): random genome, substitutions, a Q2 tail on a third of the reads, a few barcodes."""
    from superplus_amd import feudal
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, genome_size, dtype=np.uint8)
    L = read_len
    codes, pqs = [], []
    for _ in range(n_pairs):
        a = int(rng.integers(0, genome_size - 2 * L - 200))
        ins = L + int(rng.integers(0, 200))
        r1 = genome[a:a + L].copy()
        r2 = (3 - genome[a + ins:a + ins + L][::-1]).astype(np.uint8)
        if rng.random() < 0.5: r1, r2 = r2, r1
        for r in (r1, r2):
            hit = rng.random(L) < err
            r[hit] = (r[hit] + rng.integers(1, 4, int(hit.sum()))) & 3
            q = np.full(L, int(rng.choice([30, 35, 37])), np.uint8)
            q[rng.integers(0, L, 3)] = 20                                        # (mixed blocks among the constant runs)
            if rng.random() < 0.33: q[L - int(rng.integers(5, 40)):] = 2
            codes.append(r); pqs.append(np.frombuffer(feudal.pq_encode(q), np.uint8))
    n = 2 * n_pairs
    packed = feudal.pack_bases(np.stack(codes)).reshape(-1)
    read_len_a = np.full(n, L, np.uint32)
    base_off = (np.arange(n + 1, dtype=np.uint64) * np.uint64((L + 3) // 4)).astype(np.uint64)
    pq_off = np.concatenate([[0], np.cumsum([len(x) for x in pqs])]).astype(np.uint64)
    bc = np.repeat(rng.integers(0, 40, n_pairs).astype(np.int32), 2)            # 0 = unbarcoded
    return dict(packed=packed, base_off=base_off, read_len=read_len_a, pq_bytes=np.concatenate(pqs), pq_off=pq_off, bc=bc, n_reads=n)


def assert_same_solid(a, b, what=""):
    assert len(a) == len(b), f"{what}: {len(a)} vs {len(b)} solid k-mers"
    for f in ("w0", "w1", "edge_id", "count_ctx", "bc", "pad"):
        if not np.array_equal(a[f], b[f]):
            bad = np.nonzero(a[f] != b[f])[0]
            raise AssertionError(f"{what}: field {f} differs at {len(bad)} entries, first {bad[0]}: "
                                 f"{a[f][bad[0]]:#x} vs {b[f][bad[0]]:#x}")


def run_both(oracle, rs, K=48, min_qual=7, min_freq=3, min_bc=2, use_bc=True, ign_bc_below=0, **dfk_kw):
    from superplus_amd.dfk import Dfk
    bc = rs["bc"] if use_bc else None
    ref = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], bc, K=K,
                     min_qual=min_qual, min_freq=min_freq, min_bc=min_bc, ign_bc_below=ign_bc_below)
    d = Dfk(K=K, min_qual=min_qual, min_freq=min_freq, min_bc=min_bc, ign_bc_below=ign_bc_below,
            keep_pre_adjacency=True, **dfk_kw)
    d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], bc)
    return ref, d


from superplus_amd.dfk import digest_of  # noqa: E402,F401  (the numpy form of k_digest lives beside the binding)


def check_parity(ref, d):
    st = d.stats()
    assert np.array_equal(d.good_lens(), ref["good_len"]), "goodLens differ"
    assert st["n_inst"] == ref["n_inst"], (st["n_inst"], ref["n_inst"])
    assert st["n_distinct"] == ref["n_distinct"], (st["n_distinct"], ref["n_distinct"])
    assert d.solid_count() == ref["n_solid"], (d.solid_count(), ref["n_solid"])
    assert_same_solid(d.solid(pre_adjacency=True), ref["solid_pre"], "pre-adjacency (kmers.kvec view)")
    assert_same_solid(d.solid(), ref["solid"], "post-adjacency (Dict view)")
    assert np.array_equal(d.spectrum(), ref["hist"]), "spectrum differs"
    # the device digest is the one the oracle's entries give (it stands in for the entries at full size)
    assert d.digest() == digest_of(ref["solid"]), "digest (post-adjacency)"
    assert d.digest(pre_adjacency=True) == digest_of(ref["solid_pre"]), "digest (pre-adjacency)"
    return st
