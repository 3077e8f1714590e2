"""dfk_paths_digest and dfk_paths_verify (include/dfk.h) pinned where an answer exists: on the seven reference-written fixtures
the device verifier's eight counters are the ones oracle/verify_oracle.py computes from the reference's files, and the digests
are the ones those files give -- under several pass and batch geometries.  tests/test_gpu_fullsize_graph.py then uses both where
no oracle runs."""
import os

import numpy as np
import pytest

from oracle import verify_oracle
from tests.test_gpu_paths import KW
from tests.test_paths_oracle import CASES, decode_paths, load_reads
from tests.test_verify_oracle import EXPECT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case,K,npz,which", CASES)
def test_verifier_and_digests_on_reference_fixtures(golden_dir, tmp_path, monkeypatch, case, K, npz, which):
    from superplus_amd.dfk import Dfk, VERIFY
    rs = load_reads(golden_dir, which)
    kw = dict(KW[case]); nobc = kw.pop("nobc", False)
    d_fix = os.path.join(golden_dir, case)
    paths = decode_paths(open(os.path.join(d_fix, "a.paths"), "rb").read())
    _, _, _, inv = verify_oracle.load_graph_dir(d_fix, K)
    have_f4 = os.path.exists(os.path.join(d_fix, "a.dup"))
    f = lambda n: open(os.path.join(d_fix, n), "rb").read()
    want = verify_oracle.expected_check_words(paths, f("a.paths.inv") if have_f4 else None, f("a.countsb") if have_f4 else None,
                                              f("a.dup") if have_f4 else None, inv)
    for extra in (dict(), dict(passes=3, inst_per_item=1500), dict(slots=2)):
        extra = dict(extra)
        if extra.pop("slots", None):
            monkeypatch.setenv("DFK_PATH_SLOTS", "2"); monkeypatch.setenv("DFK_NO_FILTER", "1")
        d = Dfk(K=K, **kw, **extra)
        d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], None if nobc else rs["bc"])
        d.graph_build()
        d.paths_build(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"])
        got = d.paths_verify(rs["packed"], rs["base_off"], rs["read_len"])
        assert [got[k] for k in VERIFY] == EXPECT[case], (case, extra)
        if len(rs["read_len"]) % 2 == 0:
            d.paths_index_write(None); d.dups_write(None)                      # everything but the files
        ck = d.paths_digest()
        for k, v in want.items():
            if k.startswith(("INV_", "COUNTSB", "SELF", "DUP")) and len(rs["read_len"]) % 2: continue
            if k.startswith(("INV_", "COUNTSB", "SELF")) and not have_f4:
                continue                                                        # (no reference file to compare with: identities below)
            if k.startswith("DUP") and not have_f4: continue
            assert ck[k] == v, (case, extra, k)
        assert ck["EDGE_KMERS"] == ck["N_SOLID"] == d.solid_count() and ck["INV_VIOLATIONS"] == 0 and ck["N_EDGES"] == len(inv)
        if len(rs["read_len"]) % 2 == 0:
            assert ck["INV_ENTRIES"] == ck["N_PATH_EDGES"] and ck["COUNTSB_SUM"] == 2 * ck["INV_ENTRIES"] - ck["SELF_INVERSE"]
            assert ck["DUP_MARKED"] <= ck["N_PLACED"] // 2 + 1 and ck["VALID"] == 7
        d.close()


def test_reads_longer_than_255_bases(oracle, tmp_path):
    """2 x 300: a batch's scratch is addressed through ONE scan of two packed 32-bit sums (k_path_slots), whose bound on the
    batch size comes from the longest read -- not from an assumed 255 bases."""
    from oracle import graph_oracle, paths_oracle
    from superplus_amd.dfk import Dfk
    from tests import util
    rs = util.make_long_set(611, 60000, 2500, read_len=300)
    ref = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"], K=48)
    g = graph_oracle.run(ref["solid"], 48)
    reads, quals = paths_oracle.unpack_reads(rs)
    exp = paths_oracle.run(reads, quals, g, 48)["file"]
    for slots in (None, "2"):
        if slots: os.environ["DFK_PATH_SLOTS"] = slots
        try:
            d = Dfk(K=48, keep_inputs=True)
            d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
            d.graph_build(); d.paths_build()
            out = os.path.join(tmp_path, "a.paths"); d.paths_write(out)
            assert open(out, "rb").read() == exp
            v = d.paths_verify(rs["packed"], rs["base_off"], rs["read_len"])
            assert v["broken"] == 0 and v["no_anchor"] == 0 and v["dict_bad"] == 0 and v["placed"] > 4000, v
            d.close()
        finally:
            os.environ.pop("DFK_PATH_SLOTS", None)


def test_paths_build_checks_the_offset_tables_it_is_given():
    from superplus_amd.dfk import Dfk, DfkError
    from tests import util
    rs = util.make_set(612, 40000, 2000)
    d = Dfk(K=48)
    d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    d.graph_build()
    bad = rs["base_off"].copy(); bad[7] = bad[9]                                 # not monotone
    with pytest.raises(DfkError) as e:
        d.paths_build(rs["packed"], bad, rs["read_len"], rs["pq_bytes"], rs["pq_off"])
    assert e.value.code == -5
    short = rs["base_off"].copy(); short[-1] -= 9                                # the last read is left too few bytes
    with pytest.raises(DfkError):
        d.paths_build(rs["packed"], short, rs["read_len"], rs["pq_bytes"], rs["pq_off"])
    d.paths_build(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"])
    d.close()
