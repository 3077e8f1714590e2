"""SURVEY 8(f)-1 on the GPU: unipath edges on the device dictionary + canonical HyperBasevector + the graph files of
a.<K>/, byte for byte against (a) the fixtures written by the reference's own classes (tests/golden/graph_*) and
(b) the graph oracle on seeded inputs."""
import os

import numpy as np
import pytest

from tests import util
from tests.test_graph_oracle import FILES, expected_files
from tests.test_oracle_golden import load_hot, load_inputs

pytestmark = pytest.mark.gpu


def product_files(d, tmp_path, name):
    out = os.path.join(tmp_path, name)
    os.makedirs(out, exist_ok=True)
    st = d.graph_build()
    d.graph_write(out)
    return {f: open(os.path.join(out, f), "rb").read() for f in FILES}, st


def load_special(golden_dir):
    from superplus_amd import feudal
    packed, base_off, read_len = feudal.read_fastb(os.path.join(golden_dir, "special.fastb"))
    pq, pq_off = feudal.read_qualp(os.path.join(golden_dir, "special.qualp"))
    return dict(packed=packed, base_off=base_off, read_len=read_len, pq_bytes=pq, pq_off=pq_off, bc=None, n_reads=len(read_len))


@pytest.mark.parametrize("case,K,kw,which", [
    ("graph_k48", 48, dict(min_bc=2), "reads"), ("graph_k40_nobc", 40, dict(min_bc=0, nobc=True), "reads"),
    ("graph_k60_nobc", 60, dict(min_bc=0, nobc=True), "reads"), ("graph_hot_k48_minfreq2", 48, dict(min_freq=2), "hot"),
    ("graph_special_k48", 48, dict(min_bc=0, nobc=True), "special")])
def test_graph_files_match_reference_fixtures(golden_dir, tmp_path, case, K, kw, which):
    from superplus_amd.dfk import Dfk
    rs = {"reads": load_inputs, "hot": load_hot, "special": load_special}[which](golden_dir)
    kw = dict(kw)
    nobc = kw.pop("nobc", False)
    for extra in (dict(), dict(passes=3, inst_per_item=1500)):            # one part / several parts and tiny items
        d = Dfk(K=K, **kw, **extra)
        d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], None if nobc else rs["bc"])
        got, st = product_files(d, tmp_path, case + ("_p" if extra else ""))
        exp = expected_files(golden_dir, case)
        for f in FILES:
            assert got[f] == exp[f], f"{case}/{f} {extra}"
        assert st["n_edges"] == int.from_bytes(exp["a.kmers"][8:16], "little")
        d.close()


@pytest.mark.parametrize("K,seed,G,pairs,kw", [(48, 301, 60000, 3000, dict()), (48, 302, 300000, 40000, dict(passes=4)),
                                               (40, 303, 80000, 8000, dict()), (60, 304, 80000, 10000, dict())])
def test_graph_matches_oracle_on_synthetic_reads(oracle, tmp_path, K, seed, G, pairs, kw):
    """reads -> C oracle dictionary -> graph oracle, against the product's files.  (Not with min_freq = 1: that skips
    recomputeAdjacencies, BuildReadQGraph48.cc:313, contexts then name neighbours that may be absent and the reference's
    edge builder asserts on them -- not a graph input.)"""
    from oracle import graph_oracle
    from superplus_amd.dfk import Dfk
    rs = util.make_set(seed, G, pairs)
    dkw = dict(kw)
    passes = dkw.pop("passes", 0)
    ref = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"], K=K, **dkw)
    exp = graph_oracle.run(ref["solid"], K)["files"]
    d = Dfk(K=K, passes=passes, **dkw)
    d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    got, st = product_files(d, tmp_path, "g")
    for f in FILES:
        assert got[f] == exp[f], f
    # every k-mer carries its place on an edge now: (edge id, offset) as KDef::set leaves them
    s = d.solid()
    assert (s["edge_id"] != 0xFFFFFFFF).all() and st["n_canonical_edges"] == len(np.unique(s["edge_id"]))
    place = graph_oracle.run(ref["solid"], K)["place"]
    kms = graph_oracle._kmer_ints(s, K)
    per_edge = {}
    for km, e, cc in zip(kms, s["edge_id"], s["count_ctx"]):
        per_edge.setdefault(int(e), []).append((int(cc) & 0xFFFFFF, place[km]))
    for e, lst in per_edge.items():
        ref_edges = {p[1][0] for p in lst}
        assert len(ref_edges) == 1                                         # same grouping of k-mers into edges
        assert all(off == p[1] for off, p in lst)                          # same offsets
    d.close()


def test_graph_needs_a_count_and_survives_a_recount(oracle, tmp_path):
    from superplus_amd.dfk import Dfk, DfkError
    d = Dfk(K=48)
    with pytest.raises(DfkError):
        d.graph_build()
    rs = util.make_set(311, 60000, 3000)
    d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    a, _ = product_files(d, tmp_path, "a")
    d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    with pytest.raises(DfkError):
        d.graph_write(str(tmp_path))                                       # the graph belonged to the previous count
    b, _ = product_files(d, tmp_path, "b")
    assert a == b
