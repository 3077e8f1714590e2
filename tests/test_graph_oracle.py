"""The graph oracle (oracle/graph_oracle.py: unipath edges, canonical HBV, a.<K>/ files) against the fixtures the
reference's own classes produced (tests/golden/graph_*: oracle/_ref/refdrv graph = the real KmerDict and KMer walk the
edges, the real digraphE<basevector>, vecbvec and BinaryWriter write the files; oracle/ref_graph.cc).  CPU only."""
import os

import numpy as np
import pytest

from oracle import graph_oracle
from tests.test_oracle_golden import load_hot, load_inputs

FILES = ("a.k", "a.fastb", "a.hbv", "a.hbx", "a.kmers", "a.inv", "a.to_left", "a.to_right")
# (fixture dir, K, which expected dictionary)
CASES = [("graph_k48", 48, "expect_k48.npz"), ("graph_k40_nobc", 40, "expect_k40_nobc.npz"), ("graph_k60_nobc", 60, "expect_k60_nobc.npz"),
         ("graph_hot_k48_minfreq2", 48, "expect_hot_k48_minfreq2.npz"), ("graph_special_k48", 48, "expect_special_k48_nobc.npz")]


def expected_files(golden_dir, case):
    return {f: open(os.path.join(golden_dir, case, f), "rb").read() for f in FILES}


@pytest.mark.parametrize("case,K,npz", CASES)
def test_graph_oracle_matches_reference_files(golden_dir, case, K, npz):
    """From the reference's dictionary (post-recomputeAdjacencies entries) to every graph file, byte for byte."""
    solid = np.load(os.path.join(golden_dir, npz))["solid_post"]
    r = graph_oracle.run(solid, K)
    exp = expected_files(golden_dir, case)
    for f in FILES:
        assert r["files"][f] == exp[f], f"{case}/{f}"
    # every solid k-mer lies on exactly one canonical edge, at one offset
    assert len(r["place"]) == len(solid)
    assert sum(len(e) - K + 1 for e in r["edges"]) == len(solid)


def test_special_input_has_the_corner_cases(golden_dir):
    """The special fixture really contains a branch-free cycle, a palindromic one-k-mer edge and branch vertices."""
    solid = np.load(os.path.join(golden_dir, "expect_special_k48_nobc.npz"))["solid_post"]
    r = graph_oracle.run(solid, 48)
    h = r["hbv"]
    loops = [e for e, (a, b) in enumerate(zip(*h.to_left_right())) if a == b]
    assert any(len(h.edges[e]) == 400 + 47 for e in loops)                       # the 400-base circle: 400 k-mers, closed on itself
    assert any(len(s) == 48 and graph_oracle.rc_seq(s) == s for s in h.edges)    # the palindrome, an edge of its own
    assert max(len(v) for v in h.frm) >= 2                                       # a vertex with two outgoing edges
    inv = h.involution()
    assert all(inv[inv[e]] == e for e in range(len(inv)))


def test_oracle_chain_from_reads(oracle, golden_dir):
    """reads -> C oracle dictionary -> graph oracle reproduces the fixture too (the two restatements compose)."""
    rs = load_hot(golden_dir)
    r = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"], K=48, min_freq=2)
    g = graph_oracle.run(r["solid"], 48)
    exp = expected_files(golden_dir, "graph_hot_k48_minfreq2")
    for f in FILES:
        assert g["files"][f] == exp[f], f
    rs = load_inputs(golden_dir)
    r = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], None, K=60, min_freq=3, min_bc=0)
    g = graph_oracle.run(r["solid"], 60)
    exp = expected_files(golden_dir, "graph_k60_nobc")
    for f in FILES:
        assert g["files"][f] == exp[f], f
