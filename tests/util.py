"""Shared helpers for the dfk tests."""
import numpy as np
import torch

from superplus_amd import synth


def make_set(seed, genome_size, n_pairs, read_len=100, **kw):
    genome = synth.make_genome(genome_size, seed, repeat_frac=kw.pop("repeat_frac", 0.02))
    return synth.make_reads(genome, n_pairs, seed + 1, read_len=read_len, **kw).numpy()


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
