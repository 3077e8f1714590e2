"""The CPU form of the path verifier (oracle/verify_oracle.py) on the files the reference's own classes wrote
(tests/golden/graph_*/): what the device verifier (dfk_paths_verify) treats as hard invariants must hold on them -- no path with
edges that do not meet, no offset behind the first edge, no placed read without a k-mer where its path says it is -- and the
numbers it finds are the ones tests/test_gpu_verify.py expects from the device for the same inputs.  CPU only."""
import os

import pytest

from oracle import paths_oracle, verify_oracle
from tests.test_paths_oracle import CASES, decode_paths, load_reads

# [placed, broken, hits, consistent, no anchor, all consistent, dictionary bad, outside] per fixture, as computed from the
# reference-written files by oracle/verify_oracle.py (kept here so that a change of the verifier's definition shows)
EXPECT = {
    "graph_k48": [1215, 0, 35372, 35185, 0, 1211, 0, 4937],
    "graph_k40_nobc": [1409, 0, 51780, 51394, 0, 1402, 0, 3723],
    "graph_k60_nobc": [771, 0, 20003, 20003, 0, 771, 0, 3386],
    "graph_hot_k48_minfreq2": [92, 0, 1652, 1652, 0, 92, 0, 134],
    "graph_special_k48": [700, 0, 37036, 37036, 0, 700, 0, 64],
    "graph_pathy_k48": [3357, 0, 146851, 146029, 0, 3302, 0, 1947],
    "graph_frag_k48": [5634, 0, 232037, 223901, 0, 5277, 0, 15832],
    "graph_pathy2_k48": [18556, 0, 772254, 757058, 0, 17781, 0, 27569],
}


def fixture_counters(golden_dir, case, K, which):
    d = os.path.join(golden_dir, case)
    edges, left, right, _ = verify_oracle.load_graph_dir(d, K)
    paths = decode_paths(open(os.path.join(d, "a.paths"), "rb").read())
    reads, _ = paths_oracle.unpack_reads(load_reads(golden_dir, which))
    return verify_oracle.verify(edges, left, right, paths, reads, K)


@pytest.mark.parametrize("case,K,npz,which", CASES)
def test_reference_written_paths_satisfy_the_verifier(golden_dir, case, K, npz, which):
    c = fixture_counters(golden_dir, case, K, which)
    assert c == EXPECT[case]
    assert c[1] == 0 and c[4] == 0 and c[6] == 0          # the hard invariants
    assert c[3] >= 0.96 * c[2]                            # nearly every dictionary hit is where the path says


def test_verifier_sees_a_broken_path(golden_dir):
    """...and it is not blind: shift one read's offset, swap an edge for its neighbour, move a read to another's path."""
    case, K, _, which = CASES[0]
    d = os.path.join(golden_dir, case)
    edges, left, right, _ = verify_oracle.load_graph_dir(d, K)
    paths = decode_paths(open(os.path.join(d, "a.paths"), "rb").read())
    reads, _ = paths_oracle.unpack_reads(load_reads(golden_dir, which))
    placed = [i for i, (_, p) in enumerate(paths) if p]
    good = verify_oracle.verify(edges, left, right, paths, reads, K)
    i = placed[0]
    shifted = list(paths); shifted[i] = (paths[i][0] + 1, paths[i][1])
    c = verify_oracle.verify(edges, left, right, shifted, reads, K)
    assert c[3] < good[3] and c[4] == 1
    two = next(j for j in placed if len(paths[j][1]) >= 2)
    cut = list(paths); cut[two] = (paths[two][0], [paths[two][1][0], paths[two][1][0] ^ 1])
    c = verify_oracle.verify(edges, left, right, cut, reads, K)
    assert c[1] >= 1 or c[3] < good[3]


def test_digest_forms_distinguish_files(golden_dir):
    d = os.path.join(golden_dir, "graph_frag_k48")
    _, _, _, inv = verify_oracle.load_graph_dir(d, 48)
    paths = decode_paths(open(os.path.join(d, "a.paths"), "rb").read())
    f = lambda n: open(os.path.join(d, n), "rb").read()
    w = verify_oracle.expected_check_words(paths, f("a.paths.inv"), f("a.countsb"), f("a.dup"), inv)
    assert w["COUNTSB_SUM"] == 2 * w["INV_ENTRIES"] - w["SELF_INVERSE"] and w["INV_ENTRIES"] == w["N_PATH_EDGES"]
    assert 0 < w["DUP_MARKED"] <= w["N_PLACED"] // 2 + 1
    swapped = list(paths); swapped[0], swapped[1] = swapped[1], swapped[0]
    if swapped[0] != swapped[1]:
        assert verify_oracle.paths_digest(swapped) != (w["PATHS_SUM"], w["PATHS_XOR"])
