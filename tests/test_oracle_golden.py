"""The oracle (CPU restatement) against the golden vectors the reference's own classes produced
(tests/golden/make_golden.py) -- this is what pins the oracle.  CPU only."""
import os
import re
import struct

import numpy as np
import pytest

from superplus_amd import feudal
from tests import util


def load_inputs(golden_dir):
    packed, base_off, read_len = feudal.read_fastb(os.path.join(golden_dir, "reads.fastb"))
    pq, pq_off = feudal.read_qualp(os.path.join(golden_dir, "reads.qualp"))
    bci = feudal.read_bci(os.path.join(golden_dir, "reads.bci"))
    bc = feudal.bci_to_bc(bci, len(read_len))
    return dict(packed=packed, base_off=base_off, read_len=read_len, pq_bytes=pq, pq_off=pq_off, bc=bc, bci=bci,
                n_reads=len(read_len))


def load_raw(golden_dir):
    raw = open(os.path.join(golden_dir, "reads.raw"), "rb").read()
    (n,) = struct.unpack("<Q", raw[:8]); o = 8
    reads, quals = [], []
    for _ in range(n):
        (L,) = struct.unpack("<I", raw[o:o + 4]); o += 4
        reads.append(np.frombuffer(raw, np.uint8, L, o)); o += L
        quals.append(np.frombuffer(raw, np.uint8, L, o)); o += L
    return reads, quals


CASES = [("k48", 48, True, 2), ("k48_minbc1", 48, True, 1), ("k40_nobc", 40, False, 0), ("k60_nobc", 60, False, 0)]


def test_known_answers(oracle, golden_dir):
    """KMer<K> ctor / hash / isRev / rc and the KMerContext rc table, as printed by the reference."""
    txt = open(os.path.join(golden_dir, "kat.txt")).read()
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    n = 0
    for m in re.finditer(r"K=(\d+) ([ACGT]+) w0=(\w+) w1=(\w+) hash=(\w+) isRev=(\d) isPal=(\d) rc_w0=(\w+) rc_w1=(\w+) sizeof=16", txt):
        K, s = int(m.group(1)), m.group(2)
        w0, w1 = oracle.kmer_from_codes(np.array([code[c] for c in s], np.uint8), K)
        assert (w0, w1) == (int(m.group(3), 16), int(m.group(4), 16))
        assert oracle.fnv1a16(w0, w1) == int(m.group(5), 16)
        assert oracle.is_rev(w0, w1, K) == bool(int(m.group(6)))
        assert oracle.rc(w0, w1, K) == (int(m.group(8), 16), int(m.group(9), 16))
        if int(m.group(7)):
            assert oracle.rc(w0, w1, K) == (w0, w1)
        n += 1
    assert n == 4
    assert "sizeof KDef=8 Entry48=32 Entry40=32 Entry60=32 KMerContext=1" in txt
    table = [int(x, 16) for x in re.search(r"ctxrc((?: [0-9a-f]{2}){256})", txt).group(1).split()]
    assert [oracle.ctx_rc(i) for i in range(256)] == table


def test_formats_against_reference_written_files(oracle, golden_dir):
    """reads.fastb/.qualp were written by the reference's BaseVec / PQVecEncoder / feudal writer."""
    rs = load_inputs(golden_dir)
    reads, quals = load_raw(golden_dir)
    assert rs["n_reads"] == len(reads) == 1800
    assert np.array_equal(rs["read_len"], [len(r) for r in reads])
    for i in (0, 1, 17, 500, 1799):
        p = rs["packed"][int(rs["base_off"][i]):int(rs["base_off"][i + 1])]
        codes = np.array([(p[j >> 2] >> (2 * (j & 3))) & 3 for j in range(len(reads[i]))], np.uint8)
        assert np.array_equal(codes, reads[i])
    for i in range(len(reads)):
        got = oracle.pq_decode(rs["pq_bytes"][int(rs["pq_off"][i]):int(rs["pq_off"][i + 1])])
        assert np.array_equal(got, quals[i]), f"PQVec decode differs for read {i}"
    # our writers reproduce the reference-written files byte for byte
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        feudal.write_fastb(d + "/a.fastb", rs["packed"], rs["base_off"], rs["read_len"])
        feudal.write_qualp(d + "/a.qualp", rs["pq_bytes"], rs["pq_off"])
        feudal.write_bci(d + "/a.bci", rs["bci"])
        for ext in ("fastb", "qualp", "bci"):
            assert open(f"{d}/a.{ext}", "rb").read() == open(os.path.join(golden_dir, f"reads.{ext}"), "rb").read(), ext


def test_pq_encoders_roundtrip(oracle, golden_dir):
    _, quals = load_raw(golden_dir)
    for q in quals[:200]:
        assert np.array_equal(oracle.pq_decode(oracle.pq_encode(q)), q)
        assert np.array_equal(oracle.pq_decode(np.frombuffer(feudal.pq_encode(q), np.uint8)), q)
    assert list(oracle.pq_encode(np.zeros(0, np.uint8))) == [0]


@pytest.mark.parametrize("tag,K,use_bc,min_bc", CASES)
def test_oracle_matches_golden(oracle, golden_dir, tag, K, use_bc, min_bc):
    rs = load_inputs(golden_dir)
    exp = np.load(os.path.join(golden_dir, f"expect_{tag}.npz"))
    r = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"],
                   rs["bc"] if use_bc else None, K=K, min_qual=7, min_freq=3, min_bc=min_bc)
    assert np.array_equal(r["good_len"], exp["good_len"])
    util.assert_same_solid(r["solid_pre"], exp["solid_pre"], "kmers.kvec view")
    util.assert_same_solid(r["solid"], exp["solid_post"], "Dict view after recomputeAdjacencies")
    assert np.array_equal(r["hist"], exp["spectrum"])


def load_variants(golden_dir):
    import json
    return json.load(open(os.path.join(golden_dir, "variants.json")))


def check_variant(v, n_solid, digest_pre, digest_post, hist, tag):
    assert n_solid == v["n_solid"], (tag, n_solid, v["n_solid"])
    assert [str(x) for x in digest_pre] == v["digest_pre"], tag + ": kmers.kvec view"
    assert [str(x) for x in digest_post] == v["digest_post"], tag + ": Dict view after recomputeAdjacencies"
    assert [int(x) for x in hist] == v["spectrum"], tag + ": spectrum"


def test_oracle_matches_golden_filter_variants(oracle, golden_dir):
    """MIN_FREQ 1/2/5, MIN_BC 0/3/4, ignBcBelow > 0 (areIgnoredBarcodes) and the no-barcode form, each run by the
    reference's classes (refdrv dict ... ignBcBelow): solid count, spectrum and the digests of both views."""
    rs = load_inputs(golden_dir)
    for tag, v in load_variants(golden_dir).items():
        r = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"],
                       rs["bc"] if v["use_bc"] else None, K=v["K"], min_qual=7, min_freq=v["min_freq"], min_bc=v["min_bc"],
                       ign_bc_below=v["ign_bc_below"])
        check_variant(v, r["n_solid"], util.digest_of(r["solid_pre"]), util.digest_of(r["solid"]), r["hist"], tag)


def load_hot(golden_dir):
    packed, base_off, read_len = feudal.read_fastb(os.path.join(golden_dir, "hot.fastb"))
    pq, pq_off = feudal.read_qualp(os.path.join(golden_dir, "hot.qualp"))
    bci = feudal.read_bci(os.path.join(golden_dir, "hot.bci"))
    return dict(packed=packed, base_off=base_off, read_len=read_len, pq_bytes=pq, pq_off=pq_off,
                bc=feudal.bci_to_bc(bci, len(read_len)), n_reads=len(read_len))


def test_oracle_matches_golden_hot_minimizer(oracle, golden_dir):
    rs = load_hot(golden_dir)
    exp = np.load(os.path.join(golden_dir, "expect_hot_k48_minfreq2.npz"))
    r = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"], K=48, min_freq=2)
    assert np.array_equal(r["good_len"], exp["good_len"])
    util.assert_same_solid(r["solid_pre"], exp["solid_pre"], "hot: kmers.kvec view")
    util.assert_same_solid(r["solid"], exp["solid_post"], "hot: Dict view")
    assert np.array_equal(r["hist"], exp["spectrum"])


def test_good_len_rule(oracle):
    K = 48
    q = np.full(100, 30, np.uint8)
    assert oracle.good_len(q, K) == 100
    q[95:] = 2
    assert oracle.good_len(q, K) == 95
    q[40] = 2                                   # splits the good stretch: right part 41..94 (54 >= K) wins
    assert oracle.good_len(q, K) == 95
    q[60] = 2                                   # right part now 61..94 = 34 < K; 41..59 short; 0..39 short -> 0
    assert oracle.good_len(q, K) == 0
    assert oracle.good_len(np.full(47, 30, np.uint8), K) == 0
    assert oracle.good_len(np.full(48, 30, np.uint8), K) == 48
    assert oracle.good_len(np.zeros(0, np.uint8), K) == 0


def test_spectrum_json_text(oracle):
    s = oracle.spectrum_json(np.array([0, 0, 0, 5, 7], np.int64))
    assert s == ('{\n\t"description": "kmer_count",\n\t"stage": "DF",\n\t"binsize": 1,\n\t"min": 0,\n'
                 '\t"max": 4,\n\t"numbins": 5,\n\t"vals": [0,0,0,5,7]\n}\n')
    assert '"max": -1,\n\t"numbins": 0,\n\t"vals": []' in oracle.spectrum_json(np.zeros(0, np.int64))


def test_kmerize_contexts(oracle):
    """First k-mer has only a successor, last only a predecessor; a read of exactly K good bases emits nothing."""
    rng = np.random.default_rng(5)
    codes = rng.integers(0, 4, 52, dtype=np.uint8)
    packed = feudal.pack_bases(codes[None, :]).reshape(-1)
    inst = oracle.kmerize(packed, np.array([0, 13], np.uint64), np.array([52], np.uint32), None, K=48)
    assert len(inst) == 5
    for j, e in enumerate(inst):
        w = oracle.kmer_from_codes(codes[j:j + 48], 48)
        rev = oracle.is_rev(*w, 48)
        ctx = (0 if j == 0 else 0x10 << int(codes[j - 1])) | (0 if j == 4 else 1 << int(codes[j + 48]))
        assert (int(e["w0"]), int(e["w1"])) == (oracle.rc(*w, 48) if rev else w)
        assert int(e["ctx"]) == (oracle.ctx_rc(ctx) if rev else ctx)
    assert len(oracle.kmerize(packed, np.array([0, 13], np.uint64), np.array([48], np.uint32), None, K=48)) == 0
