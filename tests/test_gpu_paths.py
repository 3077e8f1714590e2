"""SURVEY 8(f)-2 on the GPU: every read pathed on the device (dfk_paths_build) and a.paths written through the ABI, byte
for byte against (a) the files the reference's own classes produced (tests/golden/graph_*/a.paths; the pathy input reaches
every rule of algorithmTwo and of the two extensions, tests/golden/pathy_rules.txt) and (b) the Python oracle on seeded reads."""
import os

import numpy as np
import pytest

from tests import util
from tests.test_paths_oracle import CASES, decode_paths, load_reads

pytestmark = pytest.mark.gpu

KW = {"graph_k48": dict(min_bc=2), "graph_k40_nobc": dict(min_bc=0, nobc=True), "graph_k60_nobc": dict(min_bc=0, nobc=True),
      "graph_hot_k48_minfreq2": dict(min_freq=2), "graph_special_k48": dict(min_bc=0, nobc=True), "graph_pathy_k48": dict(min_bc=2),
      "graph_frag_k48": dict(min_bc=2), "graph_pathy2_k48": dict(min_bc=2)}


def explain(got, exp):
    g, w = decode_paths(got), decode_paths(exp)
    bad = [i for i, (a, b) in enumerate(zip(g, w)) if a != b]
    return f"{len(bad)} of {len(w)} reads differ (sizes {len(got)} / {len(exp)}), first {bad[:5]}: got {[g[i] for i in bad[:3]]} want {[w[i] for i in bad[:3]]}"


@pytest.mark.parametrize("case,K,npz,which", CASES)
def test_paths_file_matches_reference_fixture(golden_dir, tmp_path, monkeypatch, case, K, npz, which):
    from superplus_amd.dfk import Dfk
    rs = load_reads(golden_dir, which)
    kw = dict(KW[case]); nobc = kw.pop("nobc", False)
    exp = open(os.path.join(golden_dir, case, "a.paths"), "rb").read()
    # one part / several parts; reads re-uploaded / kept; room for two parts a read (batches done again with all of it) and no
    # filter in front of the index
    # ... and a.paths streamed into its file while the reads are pathed (dfk_paths_sink)
    for extra in (dict(), dict(passes=3, inst_per_item=1500, keep_inputs=True), dict(sink=True), dict(slots=2)):
        extra = dict(extra)
        sink = extra.pop("sink", False)
        if extra.pop("slots", None):
            monkeypatch.setenv("DFK_PATH_SLOTS", "2"); monkeypatch.setenv("DFK_NO_FILTER", "1"); sink = True
        d = Dfk(K=K, **kw, **extra)
        d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], None if nobc else rs["bc"])
        d.graph_build()
        out = os.path.join(tmp_path, "a.paths")
        if sink: d.paths_sink(out)
        st = d.paths_build() if extra.get("keep_inputs") else d.paths_build(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"])
        d.paths_write(out)
        got = open(out, "rb").read()
        assert got == exp, f"{case} {extra}: " + explain(got, exp)
        want = decode_paths(exp)
        assert st["n_reads"] == len(want) and st["n_placed"] == sum(1 for _, p in want if p) and st["n_path_edges"] == sum(len(p) for _, p in want)
        off, first, edges = d.paths()
        assert [(int(o), [int(e) for e in edges[int(a):int(b)]]) for o, a, b in zip(off, first[:-1], first[1:])] == want
        d.close()


@pytest.mark.parametrize("K,seed,G,pairs,kw", [(48, 401, 60000, 4000, dict()), (48, 402, 150000, 15000, dict(passes=4)),
                                               (40, 403, 80000, 8000, dict()), (60, 404, 80000, 10000, dict())])
def test_paths_match_oracle_on_synthetic_reads(oracle, tmp_path, K, seed, G, pairs, kw):
    """reads -> C oracle dictionary -> graph oracle -> paths oracle, against the product's a.paths."""
    from oracle import graph_oracle, paths_oracle
    from superplus_amd.dfk import Dfk
    rs = util.make_set(seed, G, pairs)
    dkw = dict(kw); passes = dkw.pop("passes", 0)
    ref = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"], K=K, **dkw)
    g = graph_oracle.run(ref["solid"], K)
    reads, quals = paths_oracle.unpack_reads(rs)
    exp = paths_oracle.run(reads, quals, g, K)["file"]
    d = Dfk(K=K, passes=passes, keep_inputs=True, **dkw)
    d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    d.graph_build()
    d.paths_build()
    out = os.path.join(tmp_path, "a.paths")
    d.paths_write(out)
    got = open(out, "rb").read()
    assert got == exp, explain(got, exp)
    d.close()


def test_paths_need_a_graph(oracle):
    from superplus_amd.dfk import Dfk, DfkError
    rs = util.make_set(411, 60000, 3000)
    d = Dfk(K=48)
    d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    with pytest.raises(DfkError):
        d.paths_build(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"])     # no graph yet
    d.graph_build()
    with pytest.raises(DfkError):
        d.paths_build()                                                                             # nothing kept: keep_inputs was not set
    with pytest.raises(DfkError):
        d.paths_write("/tmp/never")                                                                 # nothing built
    d.paths_build(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"])
    d.close()


@pytest.mark.parametrize("case,which", [("graph_frag_k48", "frag"), ("graph_pathy2_k48", "pathy2")])
def test_paths_index_built_in_ranges_of_edges(golden_dir, tmp_path, monkeypatch, case, which):
    """The paths index built a range of edges at a time (what a set with 2^31 path entries or more, or short room, makes it do):
    forced here with ranges of a few hundred entries -- dozens of ranges, some of one heavy edge -- the same two files, the same
    digests as in one piece."""
    from superplus_amd.dfk import Dfk
    rs = load_reads(golden_dir, which)
    words = []
    for cap in (None, "700", "40"):
        if cap: monkeypatch.setenv("DFK_PIDX_RANGE_PAIRS", cap)
        d = Dfk(K=48, keep_inputs=True)
        d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
        d.graph_build(); d.paths_build()
        out = os.path.join(tmp_path, cap or "whole"); os.makedirs(out)
        d.paths_index_write(out)
        for f in ("a.paths.inv", "a.countsb"):
            assert open(os.path.join(out, f), "rb").read() == open(os.path.join(golden_dir, case, f), "rb").read(), f"{f} ranges of {cap}"
        ck = d.paths_digest()
        words.append({k: ck[k] for k in ("INV_SUM", "INV_XOR", "INV_STARTS", "INV_ENTRIES", "COUNTSB_DIGEST", "COUNTSB_SUM", "SELF_INVERSE")})
        d.close()
    assert words[0] == words[1] == words[2]


def test_paths_index_and_dups_match_reference_fixture(golden_dir, tmp_path):
    """Row f-4 through the ABI on the fragmented fixture (1816 edges, PCR-duplicate pairs): a.paths.inv and a.countsb
    (writePathsIndex, written on the reference side by IncrementalWriter<ULongVec> / BinaryWriter) and a.dup (MarkDups)."""
    from superplus_amd.dfk import Dfk
    rs = load_reads(golden_dir, "frag")
    for extra in (dict(), dict(passes=3, inst_per_item=1500)):
        d = Dfk(K=48, keep_inputs=True, **extra)
        d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
        d.graph_build(); d.paths_build()
        d.paths_index_write(str(tmp_path))
        marked = d.dups_write(os.path.join(tmp_path, "a.dup"))
        for f in ("a.paths.inv", "a.countsb", "a.dup"):
            assert open(os.path.join(tmp_path, f), "rb").read() == open(os.path.join(golden_dir, "graph_frag_k48", f), "rb").read(), f"{f} {extra}"
        assert marked == int(np.frombuffer(open(os.path.join(golden_dir, "graph_frag_k48", "a.dup"), "rb").read(), np.uint8, offset=16).sum())
        d.close()


@pytest.mark.parametrize("which,case", [("frag", "graph_frag_k48"), ("pathy2", "graph_pathy2_k48")])
def test_index_and_dups_in_one_call_write_the_same_files(golden_dir, tmp_path, monkeypatch, which, case):
    """dfk_paths_index_dups_write (what DF calls): a.paths.inv's lists are written by a thread of their own while the duplicates
    are marked -- or, with the index built in several ranges, the two steps simply follow each other.  Same bytes, same digests
    as the two calls; the context holds nothing more afterwards and goes on working."""
    from superplus_amd.dfk import Dfk
    rs = load_reads(golden_dir, which)
    d = Dfk(K=48, keep_inputs=True)
    d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    d.graph_build(); d.paths_build()
    d.paths_index_write(None)                                             # (gives the k-mer index back, as either call does first)
    held = d.stats()["hbm_held"]
    words = []
    for cap in (None, "900"):
        if cap: monkeypatch.setenv("DFK_PIDX_RANGE_PAIRS", cap)
        out = os.path.join(tmp_path, cap or "whole"); os.makedirs(out)
        marked = d.paths_index_dups_write(out, os.path.join(out, "a.dup"))
        for f in ("a.paths.inv", "a.countsb", "a.dup"):
            assert open(os.path.join(out, f), "rb").read() == open(os.path.join(golden_dir, case, f), "rb").read(), f"{f} ranges of {cap}"
        assert marked == int(np.frombuffer(open(os.path.join(golden_dir, case, "a.dup"), "rb").read(), np.uint8, offset=16).sum())
        words.append(d.paths_digest())
    monkeypatch.delenv("DFK_PIDX_RANGE_PAIRS")
    d.paths_index_write(None); d.dups_write(None)
    words.append(d.paths_digest())
    assert words[0] == words[1] == words[2]
    assert d.stats()["hbm_held"] == held
    with pytest.raises(Exception):
        d.paths_index_dups_write(os.path.join(tmp_path, "no", "such", "dir"), None)
    assert d.stats()["hbm_held"] == held
    d.close()


@pytest.mark.parametrize("seed,G,pairs", [(421, 20000, 8000), (422, 60000, 12000)])
def test_paths_index_and_dups_match_oracle(oracle, tmp_path, seed, G, pairs):
    """Seeded reads with duplicated pairs: the product's three files against the Python oracle's (which is pinned by the fixture)."""
    from oracle import graph_oracle, paths_oracle
    from superplus_amd.dfk import Dfk
    rs = util.make_set(seed, G, pairs)
    ref = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"], K=48)
    g = graph_oracle.run(ref["solid"], 48)
    reads, quals = paths_oracle.unpack_reads(rs)
    r = paths_oracle.run(reads, quals, g, 48)
    exp = paths_oracle.paths_index(r["paths"], g["hbv"].involution())
    exp["a.dup"] = paths_oracle.mark_dups(r["paths"], reads, quals)
    d = Dfk(K=48, keep_inputs=True)
    d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    d.graph_build(); d.paths_build()
    d.paths_index_write(str(tmp_path)); d.dups_write(os.path.join(tmp_path, "a.dup"))
    for f, b in exp.items():
        assert open(os.path.join(tmp_path, f), "rb").read() == b, f
    d.close()


def test_paths_edge_cases(oracle, tmp_path):
    """Reads shorter than K, reads with no solid k-mer at all (random bases seen once), device-resident inputs, an odd number
    of reads, and the calls out of order."""
    import torch
    from oracle import graph_oracle, paths_oracle
    from superplus_amd.dfk import Dfk, DfkError
    rs = util.make_set(431, 40000, 3000)
    # cut some reads below K and overwrite others with poly-C (no solid k-mer): rebuild the packed arrays
    reads, quals = paths_oracle.unpack_reads(rs)
    reads = [bytes(r) for r in reads]
    for i in range(0, len(reads), 97):
        reads[i] = reads[i][:30]; quals[i] = quals[i][:30]
    rng = np.random.default_rng(9)
    for i in range(5, len(reads), 131):
        reads[i] = rng.integers(0, 4, len(reads[i]), dtype=np.uint8).tobytes()
    from superplus_amd import feudal
    packed = np.concatenate([feudal.pack_bases(np.frombuffer(r, np.uint8)[None, :])[0] for r in reads])
    read_len = np.array([len(r) for r in reads], np.uint32)
    base_off = np.concatenate([[0], np.cumsum((read_len.astype(np.uint64) + 3) // 4)]).astype(np.uint64)
    pqs = [np.frombuffer(feudal.pq_encode(np.asarray(q, np.uint8)), np.uint8) for q in quals]
    pq_bytes = np.concatenate(pqs); pq_off = np.concatenate([[0], np.cumsum([len(x) for x in pqs])]).astype(np.uint64)
    ref = oracle.run(packed, base_off, read_len, pq_bytes, pq_off, rs["bc"], K=48)
    g = graph_oracle.run(ref["solid"], 48)
    exp = paths_oracle.run(reads, quals, g, 48)
    d = Dfk(K=48)
    d.count(packed, base_off, read_len, pq_bytes, pq_off, rs["bc"])
    with pytest.raises(DfkError):
        d.paths_index_write(str(tmp_path))                                 # no paths yet
    d.graph_build()
    dev = torch.device("cuda:0")
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt) if dt is not None else np.ascontiguousarray(a)).to(dev)
    packed_d = torch.cat([t(packed, None), torch.zeros(8, dtype=torch.uint8, device=dev)])      # (the stream is read as aligned words)
    st = d.paths_build_device(packed_d, t(base_off, np.int64), t(read_len, np.int32), torch.cat([t(pq_bytes, None), torch.zeros(8, dtype=torch.uint8, device=dev)]), t(pq_off, np.int64))
    assert st["n_reads"] == len(reads)
    out = os.path.join(tmp_path, "a.paths"); d.paths_write(out)
    got = open(out, "rb").read()
    assert got == exp["file"], explain(got, exp["file"])
    want = decode_paths(got)
    assert all(not want[i][1] for i in range(0, len(reads), 97)) and all(not want[i][1] for i in range(5, len(reads), 131))
    d.paths_index_write(str(tmp_path)); d.dups_write(os.path.join(tmp_path, "a.dup"))
    files = paths_oracle.paths_index(exp["paths"], g["hbv"].involution()); files["a.dup"] = paths_oracle.mark_dups(exp["paths"], reads, quals)
    for f, b in files.items():
        assert open(os.path.join(tmp_path, f), "rb").read() == b, f
    with pytest.raises(DfkError):
        d.paths_build(packed, base_off, read_len, pq_bytes, pq_off)        # the k-mer index went back to the arena with the paths index
    d.graph_build()
    d.paths_build(packed[: int(base_off[-2])], base_off[:-1], read_len[:-1], pq_bytes[: int(pq_off[-2])], pq_off[:-1])   # an odd number of reads paths fine ...
    with pytest.raises(DfkError):
        d.dups_write(os.path.join(tmp_path, "b.dup"))                      # ... but MarkDups works on pairs
    d.close()


@pytest.mark.parametrize("per_base", [False, True])
def test_quality_histogram_of_the_kept_reads(per_base):
    """dfk_qual_hist (DF's frag_reads_orig.qhist) against a count over the decoded qualities."""
    from oracle import paths_oracle
    from superplus_amd.dfk import Dfk
    rs = util.make_set(91, 40000, 3000, **(dict(ragged_frac=0.5) if per_base else {}))
    reads, quals = paths_oracle.unpack_reads(rs)
    max_len = max(len(q) for q in quals)
    want = np.zeros((2, max_len, 256), np.int64)
    for r, q in enumerate(quals):
        for pos, v in enumerate(q): want[r & 1, pos, v] += 1
    d = Dfk(K=48, keep_inputs=True)
    d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    assert np.array_equal(d.qual_hist(max_len), want)
    short = max(1, max_len - 7)                                            # positions from max_len on are left out
    assert np.array_equal(d.qual_hist(short), want[:, :short])
    d.close()
