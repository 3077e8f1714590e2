"""The read-pathing oracle (oracle/paths_oracle.py: Pather::path, HBVPather::algorithmTwo, ExtendReadPath) against the
a.paths files the reference's own classes produced (tests/golden/graph_*/a.paths: oracle/_ref/refdrv graph = the real
KmerDict::findEntry, KMer, CF<K>::isRC and bvec iterators find the parts, the real digraphE<basevector> answers the
graph queries, the real ReadPathVec feudal writer writes the file; glue restated in oracle/ref_driver.cc /
oracle/ref_graph.cc).  CPU only."""
import os

import numpy as np
import pytest

from oracle import graph_oracle, paths_oracle
from tests.test_oracle_golden import load_hot, load_inputs

# (fixture dir, K, expected dictionary, which reads)
CASES = [("graph_k48", 48, "expect_k48.npz", "reads"), ("graph_k40_nobc", 40, "expect_k40_nobc.npz", "reads"),
         ("graph_k60_nobc", 60, "expect_k60_nobc.npz", "reads"), ("graph_hot_k48_minfreq2", 48, "expect_hot_k48_minfreq2.npz", "hot"),
         ("graph_special_k48", 48, "expect_special_k48_nobc.npz", "special"), ("graph_pathy_k48", 48, "expect_pathy_k48.npz", "pathy"),
         ("graph_frag_k48", 48, "expect_frag_k48.npz", "frag"), ("graph_pathy2_k48", 48, "expect_pathy2_k48.npz", "pathy2")]


def load_named(golden_dir, name):
    from superplus_amd import feudal
    packed, base_off, read_len = feudal.read_fastb(os.path.join(golden_dir, name + ".fastb"))
    pq, pq_off = feudal.read_qualp(os.path.join(golden_dir, name + ".qualp"))
    bci = feudal.read_bci(os.path.join(golden_dir, name + ".bci"))
    return dict(packed=packed, base_off=base_off, read_len=read_len, pq_bytes=pq, pq_off=pq_off, bc=feudal.bci_to_bc(bci, len(read_len)),
                n_reads=len(read_len))


def load_reads(golden_dir, which):
    return {"reads": load_inputs, "hot": load_hot}.get(which, lambda g: load_named(g, which))(golden_dir)


def decode_paths(b):
    """a.paths bytes -> [(offset, [edges])]"""
    n = int.from_bytes(b[:4], "little")
    var_tab = int.from_bytes(b[8:16], "little")
    offs = np.frombuffer(b, "<u8", n + 1, var_tab)
    out = []
    for i in range(n):
        seg = b[int(offs[i]):int(offs[i + 1])]
        out.append((int(np.frombuffer(seg, "<i4", 1)[0]), [int(x) for x in np.frombuffer(seg, "<i4", offset=8)]))
    return out


@pytest.mark.parametrize("case,K,npz,which", CASES)
def test_paths_oracle_matches_reference_files(golden_dir, case, K, npz, which):
    solid = np.load(os.path.join(golden_dir, npz))["solid_post"]
    g = graph_oracle.run(solid, K)
    reads, quals = paths_oracle.unpack_reads(load_reads(golden_dir, which))
    r = paths_oracle.run(reads, quals, g, K)
    exp = open(os.path.join(golden_dir, case, "a.paths"), "rb").read()
    if r["file"] != exp:
        want = decode_paths(exp)
        bad = [i for i, (a, b) in enumerate(zip(r["paths"], want)) if (a[0], list(a[1])) != b]
        raise AssertionError(f"{case}: {len(bad)} reads differ, first {bad[:5]}: got {[r['paths'][i] for i in bad[:3]]} want {[want[i] for i in bad[:3]]}")


def test_pathy_fixture_reaches_every_rule(golden_dir):
    """The reference-side run printed how often each rule of algorithmTwo / the extensions fired on the pathy input."""
    rules = {}
    for line in open(os.path.join(golden_dir, "pathy_rules.txt")):
        if line.startswith("paths:   "):
            name, n = line[9:].rsplit(None, 1)
            rules[name.strip()] = int(n)
    assert len(rules) == 16 and all(v > 0 for v in rules.values()), rules
    # ... and on the larger pathy2 input at least 50 times each (three of them fired fewer than ten times on pathy)
    rules2 = {}
    for line in open(os.path.join(golden_dir, "pathy2_rules.txt")):
        if line.startswith("paths:   "):
            name, n = line[9:].rsplit(None, 1)
            rules2[name.strip()] = int(n)
    assert set(rules2) == set(rules) and all(v >= 50 for v in rules2.values()), rules2


def test_score_truncates_like_unsigned_minus_double():
    """penalty -= 0.2*penalty on an unsigned (ExtendReadPath.cc:50): 1 -> 0, 5 -> 4, 6 -> 4, 37 -> 29."""
    read = bytes([0] * 20)
    edge = bytes([1] + [0] * 60)                       # one mismatch at the first compared base, then matches
    for q0, after in ((1, 0), (5, 4), (6, 4), (37, 29)):
        q = np.array([q0] + [30] * 19, np.uint8)
        # right overlap from read index 0: start = 20; K = 2 so that the edge is compared from its index 1... use K = 1
        s1 = paths_oracle.Pather.score(read, q, 20, edge, 1, False)
        assert s1 == q0                                 # the mismatch costs its quality; matches cost nothing
        # a second mismatch right after one matching base adds (decayed penalty + its own quality)
        edge2 = bytes([1, 0, 1] + [0] * 60)
        q2 = np.array([q0, 30, 10] + [30] * 17, np.uint8)
        assert paths_oracle.Pather.score(read, q2, 20, edge2, 1, False) == q0 + after + 10


def test_paths_index_and_dups_of_the_larger_pather_input(golden_dir):
    """The same three files for pathy2 (3184 edges, 30995 path entries, reads that cross an edge twice)."""
    solid = np.load(os.path.join(golden_dir, "expect_pathy2_k48.npz"))["solid_post"]
    g = graph_oracle.run(solid, 48)
    reads, quals = paths_oracle.unpack_reads(load_reads(golden_dir, "pathy2"))
    r = paths_oracle.run(reads, quals, g, 48)
    files = paths_oracle.paths_index(r["paths"], g["hbv"].involution())
    files["a.dup"] = paths_oracle.mark_dups(r["paths"], reads, quals)
    for f, b in files.items():
        assert b == open(os.path.join(golden_dir, "graph_pathy2_k48", f), "rb").read(), f


def test_paths_index_and_dups_match_reference_files(golden_dir):
    """Row f-4 on the fragmented fixture (1816 edges: the reference's own writePathsIndex needs more than ~870): a.paths.inv
    and a.countsb as IncrementalWriter<ULongVec> / BinaryWriter wrote them, a.dup as BinaryWriter wrote it."""
    solid = np.load(os.path.join(golden_dir, "expect_frag_k48.npz"))["solid_post"]
    g = graph_oracle.run(solid, 48)
    reads, quals = paths_oracle.unpack_reads(load_reads(golden_dir, "frag"))
    r = paths_oracle.run(reads, quals, g, 48)
    inv = g["hbv"].involution()
    assert len(inv) >= 870
    files = paths_oracle.paths_index(r["paths"], inv)
    files["a.dup"] = paths_oracle.mark_dups(r["paths"], reads, quals)
    for f, b in files.items():
        assert b == open(os.path.join(golden_dir, "graph_frag_k48", f), "rb").read(), f
    d = np.frombuffer(files["a.dup"], np.uint8, offset=16)
    assert 100 < d.sum() < len(d) // 2                         # the fixture really holds duplicate pairs
