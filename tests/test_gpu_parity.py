"""Parity of the HIP path (through the C ABI) against the oracle.  Bit-exact: integer work."""
import os

import numpy as np
import pytest

from superplus_amd import synth
from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K,seed,G,pairs", [(48, 11, 60000, 3000), (48, 12, 200000, 40000), (40, 13, 80000, 8000),
                                            (60, 14, 80000, 10000)])
def test_parity_synthetic(oracle, K, seed, G, pairs):
    rs = util.make_set(seed, G, pairs)
    ref, d = util.run_both(oracle, rs, K=K)
    st = util.check_parity(ref, d)
    assert st["n_solid"] > 0
    assert d.spectrum_json() == oracle.spectrum_json(ref["hist"])


def test_unsorted_fetch_is_a_permutation_of_the_sorted_one(oracle):
    rs = util.make_set(5, 400000, 60000)
    ref, d = util.run_both(oracle, rs, K=48, passes=2)
    a, b = d.solid(), d.solid_unsorted()
    assert len(a) == len(b) == ref["n_solid"] > 1 << 16                  # large enough for the threaded host sort
    order = np.lexsort((b["w1"], b["w0"]))
    util.assert_same_solid(b[order], a, "unsorted fetch, sorted here")
    util.assert_same_solid(a, ref["solid"], "sorted fetch")


def test_golden_fixtures_through_abi(oracle, golden_dir):
    """The committed golden vectors (reference classes, tests/golden/make_golden.py) through libdfk."""
    import os
    from superplus_amd.dfk import Dfk
    from tests.test_oracle_golden import CASES, load_inputs
    rs = load_inputs(golden_dir)
    for tag, K, use_bc, min_bc in CASES:
        exp = np.load(os.path.join(golden_dir, f"expect_{tag}.npz"))
        d = Dfk(K=K, min_bc=min_bc, keep_pre_adjacency=True)
        d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"] if use_bc else None)
        assert np.array_equal(d.good_lens(), exp["good_len"]), tag
        util.assert_same_solid(d.solid(pre_adjacency=True), exp["solid_pre"], tag + " kvec view")
        util.assert_same_solid(d.solid(), exp["solid_post"], tag + " Dict view")
        assert np.array_equal(d.spectrum(), exp["spectrum"]), tag
        d.close()


def test_golden_filter_variants_through_abi(golden_dir):
    """Every filter parameter the tests below vary, against fixtures the reference's classes produced
    (tests/golden/variants.json: refdrv dict with MIN_FREQ 1/2/5, MIN_BC 0/3/4, ignBcBelow > 0, no barcodes)."""
    from superplus_amd.dfk import Dfk
    from tests.test_oracle_golden import check_variant, load_inputs, load_variants
    rs = load_inputs(golden_dir)
    for tag, v in load_variants(golden_dir).items():
        d = Dfk(K=v["K"], min_freq=v["min_freq"], min_bc=v["min_bc"], ign_bc_below=v["ign_bc_below"], keep_pre_adjacency=True)
        d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"] if v["use_bc"] else None)
        check_variant(v, d.solid_count(), d.digest(pre_adjacency=True), d.digest(), d.spectrum(), tag)
        check_variant(v, d.solid_count(), util.digest_of(d.solid(pre_adjacency=True)), util.digest_of(d.solid()), d.spectrum(), tag + " (fetched)")
        d.close()


@pytest.mark.parametrize("kw", [dict(), dict(inst_per_item=1000), dict(passes=3)])
def test_golden_hot_minimizer_through_abi(golden_dir, kw):
    """One fine bucket holding thousands of distinct k-mers (every read carries the rank-0 minimizer) beside a diverged
    repeat family: expected entries from the reference's classes.  Goes through the sub-pass / HBM-table paths."""
    from superplus_amd.dfk import Dfk
    from tests.test_oracle_golden import load_hot
    import os
    rs = load_hot(golden_dir)
    exp = np.load(os.path.join(golden_dir, "expect_hot_k48_minfreq2.npz"))
    d = Dfk(K=48, min_freq=2, keep_pre_adjacency=True, **kw)
    d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    assert np.array_equal(d.good_lens(), exp["good_len"])
    util.assert_same_solid(d.solid(pre_adjacency=True), exp["solid_pre"], "hot: kvec view")
    util.assert_same_solid(d.solid(), exp["solid_post"], "hot: Dict view")
    assert np.array_equal(d.spectrum(), exp["spectrum"])
    assert d.stats()["n_overflow_items"] > 0
    d.close()


@pytest.mark.parametrize("min_freq,min_bc,use_bc,ign", [(1, 2, True, 0), (2, 0, True, 0), (3, 1, True, 0), (3, 2, False, 0),
                                                       (3, 2, True, 2000), (5, 2, True, 10**9),
                                                       (3, 3, True, 0), (3, 4, True, 0), (2, 3, True, 2000)])
def test_filter_variants(oracle, min_freq, min_bc, use_bc, ign):
    """MIN_FREQ / MIN_BC / no-barcode / ignBcBelow variants (areIgnoredBarcodes, areEnoughBarcodes;
    min_freq == 1 skips recomputeAdjacencies, BuildReadQGraph48.cc:313)."""
    rs = util.make_set(21, 50000, 5000)
    ref, d = util.run_both(oracle, rs, K=48, min_freq=min_freq, min_bc=min_bc, use_bc=use_bc, ign_bc_below=ign)
    util.check_parity(ref, d)


@pytest.mark.parametrize("K,min_bc", [(48, 5), (48, 8), (60, 6), (40, 7)])
def test_min_bc_up_to_eight(oracle, K, min_bc):
    """MIN_BC 5..8: a table slot remembers MIN_BC-1 distinct barcodes, one word each (up to 7 extra words per slot:
    one workgroup per CU at K=60); many barcodes per locus so that the filter bites without emptying the set."""
    rs = util.make_set(29, 60000, 9000, pairs_per_barcode=3)
    ref, d = util.run_both(oracle, rs, K=K, min_bc=min_bc)
    st = util.check_parity(ref, d)
    assert 0 < st["n_solid"] < util.run_both(oracle, rs, K=K, min_bc=2)[0]["n_solid"]
    ref, d = util.run_both(oracle, rs, K=K, min_bc=min_bc, inst_per_item=100000, passes=2)
    assert util.check_parity(ref, d)["n_overflow_items"] > 0


@pytest.mark.parametrize("K,min_bc", [(40, 3), (60, 4)])
def test_min_bc_above_two_other_k(oracle, K, min_bc):
    """MIN_BC 3 and 4 (a table slot remembers MIN_BC-1 barcodes) with 3 and 4 key words per slot; also through the
    HBM-table fallback (tiny items force overflow)."""
    rs = util.make_set(23, 60000, 6000)
    ref, d = util.run_both(oracle, rs, K=K, min_bc=min_bc)
    util.check_parity(ref, d)
    n_solid = ref["n_solid"]
    assert 0 < n_solid < util.run_both(oracle, rs, K=K, min_bc=2)[0]["n_solid"]      # the filter bites
    ref, d = util.run_both(oracle, rs, K=K, min_bc=min_bc, inst_per_item=100000)
    st = util.check_parity(ref, d)
    assert st["n_overflow_items"] > 0


def _custom(reads, quals, bc=None):
    from superplus_amd import feudal
    packed, boff, pq, poff = [], [0], [], [0]
    for r, q in zip(reads, quals):
        p = feudal.pack_bases(np.asarray(r, np.uint8)[None, :]).reshape(-1) if len(r) else np.zeros(0, np.uint8)
        packed.append(p); boff.append(boff[-1] + len(p))
        e = np.frombuffer(feudal.pq_encode(q), np.uint8)
        pq.append(e); poff.append(poff[-1] + len(e))
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.uint8)
    return dict(packed=cat(packed), base_off=np.array(boff, np.uint64), read_len=np.array([len(r) for r in reads], np.uint32),
                pq_bytes=cat(pq), pq_off=np.array(poff, np.uint64),
                bc=np.zeros(len(reads), np.int32) if bc is None else np.asarray(bc, np.int32), n_reads=len(reads))


def test_ragged_and_degenerate_reads(oracle):
    """Empty reads, reads shorter than K, exactly K (emit nothing, :153), K+1, long reads, low-quality islands."""
    rng = np.random.default_rng(3)
    g = rng.integers(0, 4, 3000, dtype=np.uint8)
    reads, quals, bc = [], [], []
    for i in range(1200):
        L = int(rng.choice([0, 1, 20, 47, 48, 49, 50, 100, 150, 251, 400]))
        pos = int(rng.integers(0, 3000 - 400))
        r = g[pos:pos + L].copy()
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.uint8)
        q = rng.choice([40, 30, 20, 8, 7, 6, 2], L, p=[.3, .3, .2, .1, .04, .03, .03]).astype(np.uint8)
        reads.append(r); quals.append(q); bc.append(int(rng.integers(0, 5)))
    rs = _custom(reads, quals, bc)
    for K in (40, 48, 60):
        ref, d = util.run_both(oracle, rs, K=K)
        util.check_parity(ref, d)
    # ... and with the bases uploaded in pieces of a few reads, the scan following them (run_under_upload)
    import os
    os.environ["DFK_SCAN_UNDER_UPLOAD_MIN"] = "0"; os.environ["DFK_UPLOAD_SEGMENT"] = "256"
    try:
        for K in (40, 48, 60):
            ref, d = util.run_both(oracle, rs, K=K)
            util.check_parity(ref, d)
    finally:
        del os.environ["DFK_SCAN_UNDER_UPLOAD_MIN"], os.environ["DFK_UPLOAD_SEGMENT"]


@pytest.mark.parametrize("nbits", [1, 2, 3, 4, 5, 6])
def test_quality_fields_of_every_width(oracle, nbits):
    """k_trim decodes per-base quality fields 64 bits at a time: every field width a PQVec block can have, read
    lengths that put the fields at every bit offset of the 8-byte windows, thresholds that cut inside the field range."""
    rng = np.random.default_rng(100 + nbits)
    g = rng.integers(0, 4, 4000, dtype=np.uint8)
    lo = 7 - (1 << nbits) // 2 if (1 << nbits) // 2 <= 7 else 0       # MIN_QUAL = 7 falls inside [lo, lo + 2^nbits)
    reads, quals = [], []
    for i in range(1500):
        L = int(rng.integers(1, 330))
        pos = int(rng.integers(0, 4000 - 330))
        q = (lo + rng.integers(0, 1 << nbits, L)).astype(np.uint8)
        q[rng.random(L) < 0.8] = lo + (1 << nbits) - 1                 # mostly good, so that runs of K good bases exist
        if L > 2:
            q[0], q[1] = lo, lo + (1 << nbits) - 1                       # the block's range is exactly nbits wide
        reads.append(g[pos:pos + L].copy()); quals.append(q)
    rs = _custom(reads, quals, None)
    from superplus_amd import feudal
    widths = set()
    for q in quals[:50]:
        e = feudal.pq_encode(q); p = 0
        while e[p]:
            widths.add(e[p + 1] & 7); p += ((e[p] * (e[p + 1] & 7) + 24) >> 3)
    assert nbits in widths
    ref, d = util.run_both(oracle, rs, K=48, min_freq=1, use_bc=False)
    st = util.check_parity(ref, d)
    assert st["n_solid"] > 0 and int(np.count_nonzero(ref["good_len"] != rs["read_len"])) > 100    # (the trim did cut reads)


def test_long_reads_take_the_scanning_scatter(oracle):
    """240-base reads have more than the ten runs a summary holds: most of them go through the scanning
    scatter (k_partition<K,true>) in every pass, and its words through several refills of the scan's LDS ring."""
    rs = util.make_set(81, 300000, 6000, read_len=240)
    for passes in (0, 4):
        ref, d = util.run_both(oracle, rs, K=48, passes=passes)
        util.check_parity(ref, d)


def test_low_complexity_hot_buckets(oracle):
    """Poly-A / dinucleotide repeats: one k-mer seen thousands of times, runs longer than NK_MAX in one bucket,
    palindromes (not context-symmetrised)."""
    rng = np.random.default_rng(4)
    reads, quals, bc = [], [], []
    pal = np.array([0, 1, 2, 3] * 12 + [0, 1, 2, 3][::-1] * 0, np.uint8)  # ACGT repeat: reverse-complement palindromic 48-mers
    for i in range(3000):
        kind = i % 4
        if kind == 0:
            r = np.zeros(100, np.uint8)
        elif kind == 1:
            r = np.tile(np.array([0, 3], np.uint8), 50)
        elif kind == 2:
            r = np.tile(np.array([0, 1, 2, 3], np.uint8), 25)
        else:
            r = rng.integers(0, 4, 100, dtype=np.uint8)
        reads.append(r); quals.append(np.full(100, 35, np.uint8)); bc.append(1 + i % 7)
    rs = _custom(reads, quals, bc)
    ref, d = util.run_both(oracle, rs, K=48)
    st = util.check_parity(ref, d)
    assert ref["hist"].size > 1000        # a bin beyond the LDS spectrum range is exercised


def test_lds_table_overflow_falls_back(oracle):
    """Nearly every k-mer distinct and items far over the LDS table's capacity: the split/HBM-table
    fallback must give the same answer."""
    rs = util.make_set(31, 3000000, 20000, err=0.0)     # ~1.3x coverage: distinct ~= instances
    ref, d = util.run_both(oracle, rs, K=48, min_freq=1, inst_per_item=60000)
    st = util.check_parity(ref, d)
    assert st["n_overflow_items"] > 0


def test_no_good_bases_is_an_error(oracle, upload_mode):
    from superplus_amd.dfk import Dfk, DfkError
    rs = _custom([np.zeros(100, np.uint8)] * 10, [np.full(100, 2, np.uint8)] * 10)
    d = Dfk(K=48)
    with pytest.raises(DfkError) as e:
        d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    assert e.value.code == -7
    with pytest.raises(DfkError):                       # n_reads == 0 is the same condition (nKmers == 0)
        d.count(np.zeros(0, np.uint8), np.zeros(1, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint8),
                np.zeros(1, np.uint64), None)


@pytest.fixture(params=["upload_then_count", "scan_under_the_upload"])
def upload_mode(request, monkeypatch):
    """dfk_count from host arrays both ways: everything uploaded first, or the bases in pieces with the scan behind them
    (the default from 1 GB of bases on; forced here with pieces of 512 bytes)."""
    if request.param == "scan_under_the_upload":
        monkeypatch.setenv("DFK_SCAN_UNDER_UPLOAD_MIN", "0"); monkeypatch.setenv("DFK_UPLOAD_SEGMENT", "512")
    return request.param


def test_malformed_pqvec_is_an_error(upload_mode):
    from superplus_amd.dfk import Dfk, DfkError
    rs = _custom([np.zeros(100, np.uint8)] * 4, [np.full(100, 30, np.uint8)] * 4)
    rs["read_len"][2] = 99
    with pytest.raises(DfkError) as e:
        Dfk(K=48).count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    assert e.value.code == -5


def test_result_independent_of_partition_geometry(oracle):
    """Size-independent property at a size the oracle is not run on: the sorted solid set, spectrum and
    counts cannot depend on minimizer length or item size (different bucketings of the same multiset)."""
    import hashlib
    from superplus_amd.dfk import Dfk
    rs = util.make_set(41, 2000000, 300000)
    sigs = []
    for M, ipi in ((14, 0), (10, 3000), (16, 9000)):
        d = Dfk(K=48, minimizer_len=M, inst_per_item=ipi)
        d.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
        s = d.solid(); h = d.spectrum(); st = d.stats()
        assert h.sum() == len(s) == st["n_solid"] and (h * np.arange(len(h))).sum() <= st["n_inst"]
        assert np.all(np.diff(s["w0"].astype(np.float64)) >= 0)
        sigs.append((hashlib.sha256(s.tobytes()).hexdigest(), hashlib.sha256(h.tobytes()).hexdigest(), st["n_distinct"]))
        d.close()
    assert sigs[0] == sigs[1] == sigs[2]


def test_single_bucket_overflow_uses_hbm_table(oracle):
    """Tens of thousands of distinct k-mers that all contain AGTACGGTATGCTCAC -- the one 16-mer whose minimizer
    hash is 0 (dfk_device.h: MMER_SALT) -- land in ONE fine bucket; it cannot be split, so it is counted in an
    HBM table (k_big_insert and friends)."""
    rng = np.random.default_rng(9)
    hot = np.array(["ACGT".index(ch) for ch in "AGTACGGTATGCTCAC"], np.uint8)
    reads, quals, bc = [], [], []
    for i in range(3000):
        r = rng.integers(0, 4, 100, dtype=np.uint8)
        r[42:58] = hot
        if i % 3 == 0:                          # some exact duplicates so that solid k-mers exist
            r = reads[-1].copy() if reads else r
        reads.append(r); quals.append(np.full(100, 30, np.uint8)); bc.append(1 + i % 5)
    rs = _custom(reads, quals, bc)
    ref, d = util.run_both(oracle, rs, K=48, min_freq=2)
    st = util.check_parity(ref, d)
    assert st["n_overflow_items"] > 0 and st["n_solid"] > 0


@pytest.mark.parametrize("passes", [2, 8])
def test_hash_slice_passes(oracle, passes):
    """Capacity passes (MapReduceEngine's nPasses idea): fine buckets are counted in slices; the union must
    equal the single-pass result and the oracle."""
    rs = util.make_set(61, 150000, 20000)
    ref, d = util.run_both(oracle, rs, K=48, passes=passes)
    st = util.check_parity(ref, d)
    assert st["n_passes"] == passes


@pytest.mark.parametrize("K,segment", [(48, 4096), (48, 1 << 20), (40, 50000), (60, 50000)])
def test_counting_scan_under_the_upload_of_the_bases(oracle, monkeypatch, K, segment):
    """dfk_count from host arrays (what DF calls): the bases go to the device in pieces, the trim runs beside the first one and
    the counting scan takes the reads of every piece as it arrives (run_under_upload) -- the default from 1 GB of bases on,
    forced here on a small ragged set with pieces of a few reads up to all of them.  Same dictionary as the oracle's, same
    digest as the run that uploads first and counts afterwards."""
    monkeypatch.setenv("DFK_SCAN_UNDER_UPLOAD_MIN", "0")
    monkeypatch.setenv("DFK_UPLOAD_SEGMENT", str(segment))
    monkeypatch.setenv("DFK_TRACE", "1")
    rs = util.make_set(91 + K, 120000, 30000)
    ref, d = util.run_both(oracle, rs, K=K, passes=2)
    st = util.check_parity(ref, d)
    under = d.digest()
    assert st["ms_part_count"] > 0 and st["n_records"] > 0
    assert st["n_scan_launches"] >= (2 if segment < len(rs["packed"]) else 1)
    del d
    monkeypatch.setenv("DFK_NO_SCAN_UNDER_UPLOAD", "1")
    ref, d = util.run_both(oracle, rs, K=K, passes=2)
    st2 = util.check_parity(ref, d)
    assert st2["n_records"] == st["n_records"] and d.digest() == under and st2["n_scan_launches"] == 1


def test_repeat_family_hot_minimizers(oracle):
    """An Alu-like family (many diverged copies of one element): a few minimizers own tens of thousands of
    distinct k-mers each, far beyond an LDS table.  Those buckets are counted in HBM tables by the whole
    grid (k_big_insert / k_big_flags / k_big_resolve / k_big_emit)."""
    genome = synth.make_genome(1_500_000, 77, family_copies=1500)
    rs = synth.make_reads(genome, 225_000, 78).numpy()
    ref, d = util.run_both(oracle, rs, K=48)
    st = util.check_parity(ref, d)
    assert st["n_overflow_items"] > 20


def test_small_hbm_budget_plans_its_own_passes(oracle):
    """With little HBM the library cuts the bucket space into ranges by itself (two passes in flight, each in its
    own block, dictionary parts reserved where the dictionary grows): same answer, more than one pass."""
    from superplus_amd.dfk import Dfk, DfkError
    rs = util.make_set(71, 5_000_000, 800_000)
    ref, d = util.run_both(oracle, rs, K=48)
    st = util.check_parity(ref, d)
    assert st["n_passes"] == 1
    budget = st["hbm_bytes_peak"] - int(0.35 * 32 * st["n_records"])    # the dictionary fits, the records do not all at once
    del d
    ref, d = util.run_both(oracle, rs, K=48, hbm_budget_bytes=budget)
    st = util.check_parity(ref, d)
    assert st["n_passes"] >= 2, st["n_passes"]
    del d
    # and a budget that cannot even hold the spectrum bins beside the reads' summaries is an error
    tiny = Dfk(K=48, hbm_budget_bytes=300 << 20)
    with pytest.raises(DfkError) as e:
        tiny.count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
    assert e.value.code == -4


def test_output_segments_too_small_are_redone(oracle):
    """MIN_FREQ=1 at low coverage: nearly every instance is a solid k-mer, far more than the first pass's prior
    (instances/16) sizes the output segments for; the pass is undone and redone with more room."""
    rs = util.make_set(73, 12_000_000, 150_000, err=0.0)
    ref, d = util.run_both(oracle, rs, K=48, min_freq=1, min_bc=0)
    st = util.check_parity(ref, d)
    assert st["n_solid"] > 8_000_000


def test_budget_above_free_hbm_is_clamped(oracle):
    """runall.sh hands every binary MAX_MEM_GB=640; a caller that forwards such a figure as the device budget must get
    90 % of the free HBM, not a plan for room that does not exist (dfk_create clamps)."""
    import torch
    rs = util.make_set(91, 200000, 30000)
    ref, d = util.run_both(oracle, rs, K=48, hbm_budget_bytes=640 << 30)
    st = util.check_parity(ref, d)
    assert st["n_passes"] == 1 and st["hbm_bytes_peak"] < torch.cuda.get_device_properties(0).total_memory


def test_count_saturates_at_2_pow_24_in_an_lds_table(oracle):
    """KDef::setCount saturates at 2^24-1 (ReadPather.h:128-129).  330 k poly-A reads put 17.5 M instances of ONE 48-mer
    into one fine bucket -- one work item, one LDS table, however many instances: the count must stop at exactly 2^24-1
    and must not carry into the slot's fingerprint byte (which makes the k-mer claim a second slot: seen in two runs
    of six before the adds near saturation were made exact)."""
    n = 330_000
    rng = np.random.default_rng(17)
    extra = rng.integers(0, 4, (3000, 100), dtype=np.uint8)
    extra[1000:2000] = extra[:1000]; extra[2000:] = extra[:1000]      # some other solid k-mers (three copies, barcodes differ)
    from superplus_amd import feudal
    packed = np.concatenate([np.zeros(25 * n, np.uint8), feudal.pack_bases(extra).reshape(-1)])
    N = n + len(extra)
    blk = np.array([100, (35 << 3) & 0xFF, 35 >> 5, 0], np.uint8)      # one nBits=0 block of Q35 + terminator
    rs = dict(packed=packed, base_off=(np.arange(N + 1, dtype=np.uint64) * 25), read_len=np.full(N, 100, np.uint32),
              pq_bytes=np.tile(blk, N), pq_off=(np.arange(N + 1, dtype=np.uint64) * 4),
              bc=(1 + np.arange(N) % 7).astype(np.int32), n_reads=N)
    ref, d = util.run_both(oracle, rs, K=48)
    st = util.check_parity(ref, d)
    s = d.solid()
    top = s[s["w0"] == 0]
    assert len(top) == 1 and (top["count_ctx"][0] & 0xFFFFFF) == 0xFFFFFF      # poly-A, saturated
    assert st["n_inst"] > (1 << 24) and len(d.spectrum()) == 1 << 24


def test_malformed_offset_tables_are_an_error(upload_mode):
    """The offset tables are checked on the device before anything is read through them (DFK_E_INPUT, no stray read)."""
    from superplus_amd.dfk import Dfk, DfkError
    base = _custom([np.zeros(100, np.uint8)] * 64, [np.full(100, 30, np.uint8)] * 64)
    def broken(key, idx, val):
        rs = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in base.items()}
        rs[key][idx] = val
        return rs
    cases = [broken("pq_off", 10, 1 << 40),                      # a quality stream far outside the array
             broken("pq_off", 10, 0),                            # not ascending
             broken("base_off", 20, 1 << 40), broken("base_off", 20, 3),
             broken("read_len", 5, 4000)]                        # more bases than the read has bytes
    for rs in cases:
        with pytest.raises(DfkError) as e:
            Dfk(K=48).count(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"])
        assert e.value.code == -5, e.value


def test_kvec_file_device_order_and_sorted(oracle, tmp_path):
    rs = util.make_set(93, 300000, 50000)
    ref, d = util.run_both(oracle, rs, K=48, passes=3)
    from superplus_amd.dfk import ENTRY_DTYPE
    def load(path):
        kv = open(path, "rb").read()
        n = int.from_bytes(kv[8:16], "little")
        assert kv[:8] == b"BINWRITE" and len(kv) == 16 + 32 * n
        return np.frombuffer(kv, ENTRY_DTYPE, count=n, offset=16)
    d.write_kvec(f"{tmp_path}/a.kvec")
    a = load(f"{tmp_path}/a.kvec")
    util.assert_same_solid(a[np.lexsort((a["w1"], a["w0"]))], ref["solid"], "device-order kvec")
    assert np.array_equal(a, d.solid_unsorted())
    d.write_kvec(f"{tmp_path}/b.kvec", in_order=True)
    util.assert_same_solid(load(f"{tmp_path}/b.kvec"), ref["solid"], "sorted kvec")
    d.write_kvec(f"{tmp_path}/c.kvec", pre_adjacency=True)
    c = load(f"{tmp_path}/c.kvec")
    util.assert_same_solid(c[np.lexsort((c["w1"], c["w0"]))], ref["solid_pre"], "pre-adjacency kvec")


def test_counts_beyond_2_pow_24_in_an_hbm_table(oracle):
    """The same limit where the whole grid shares a table.  4.3 M copies of one read made of the 16-mer whose minimizer
    rank is 0, repeated: 16 distinct 48-mers (the rotations), five of them with 4 x 4.3 M = 1.72x10^7 instances
    (saturated: exactly 2^24-1), eleven with 3 x 4.3 M = 1.29x10^7 (exact); 3000 random reads around the same 16-mer
    put enough distinct k-mers into the SAME fine bucket that it goes to the HBM-table path.  Expected values are
    analytic for the periodic k-mers and the oracle's for the rest (the two sets of k-mers are disjoint)."""
    from superplus_amd import feudal
    from superplus_amd.dfk import Dfk
    N = 4_300_000
    hot = np.array(["ACGT".index(ch) for ch in "AGTACGGTATGCTCAC"], np.uint8)
    periodic = np.tile(hot, 7)[:100]
    rng = np.random.default_rng(23)
    rnd = rng.integers(0, 4, (3000, 100), dtype=np.uint8)
    rnd[:, 42:58] = hot
    rnd[1000:2000] = rnd[:1000]; rnd[2000:] = rnd[:1000]
    blk = np.array([100, (35 << 3) & 0xFF, 35 >> 5, 0], np.uint8)
    def as_set(bases, first_bc):
        n = len(bases)
        return dict(packed=feudal.pack_bases(bases).reshape(-1), base_off=np.arange(n + 1, dtype=np.uint64) * 25,
                    read_len=np.full(n, 100, np.uint32), pq_bytes=np.tile(blk, n), pq_off=np.arange(n + 1, dtype=np.uint64) * 4,
                    bc=(first_bc + np.arange(n) % 7).astype(np.int32))
    small = as_set(rnd, 1)
    ref = oracle.run(small["packed"], small["base_off"], small["read_len"], small["pq_bytes"], small["pq_off"], small["bc"], K=48)
    one = feudal.pack_bases(periodic[None, :]).reshape(-1)
    M = N + len(rnd)
    packed = np.concatenate([np.tile(one, N), small["packed"]])
    bc = np.concatenate([(1 + np.arange(N) % 7).astype(np.int32), small["bc"]])
    d = Dfk(K=48)
    d.count(packed, np.arange(M + 1, dtype=np.uint64) * 25, np.full(M, 100, np.uint32), np.tile(blk, M), np.arange(M + 1, dtype=np.uint64) * 4, bc)
    st = d.stats()
    assert st["n_overflow_items"] > 0                                  # the bucket left the LDS path
    assert st["n_distinct"] == ref["n_distinct"] + 16 and st["n_solid"] == ref["n_solid"] + 16
    s = d.solid()
    per = {}
    for r in range(16):
        w0, w1 = oracle.kmer_from_codes(np.tile(hot, 5)[r:r + 48], 48)
        if oracle.is_rev(w0, w1, 48):
            w0, w1 = oracle.rc(w0, w1, 48)
        per[(w0, w1)] = min(N * (4 if r <= 4 else 3), (1 << 24) - 1)
    assert len(per) == 16
    keys = list(zip(s["w0"].tolist(), s["w1"].tolist()))
    assert len(set(keys)) == len(keys)                                  # no k-mer twice
    got = {k: int(c) & 0xFFFFFF for k, c in zip(keys, s["count_ctx"]) if k in per}
    assert got == per
    rest = s[[k not in per for k in keys]]
    assert np.array_equal(rest["w0"], ref["solid"]["w0"]) and np.array_equal(rest["w1"], ref["solid"]["w1"])
    assert np.array_equal(rest["count_ctx"] & 0xFFFFFF, ref["solid"]["count_ctx"] & 0xFFFFFF)
    d.close()


def test_hinted_file_ranges_are_read_from_the_file(oracle, tmp_path):
    """dfk_hint_file_range: the arrays are maps of a file whose CONTENT is right while the mapped bytes the library is handed
    are not (a private, scribbled-over mapping) -- the count must come out of the file."""
    import mmap
    from superplus_amd.dfk import Dfk
    rs = util.make_set(77, 30000, 1500)
    ref = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"], K=48)
    names = ("packed", "base_off", "read_len", "pq_bytes", "pq_off")
    path = os.path.join(tmp_path, "arrays.bin")
    at, where = 0, {}
    with open(path, "wb") as f:
        for k in names:
            b = np.ascontiguousarray(rs[k]).tobytes()
            f.write(b); where[k] = (at, len(b)); at += len(b)
            pad = -at % 4096; f.write(b"\0" * pad); at += pad
    fd = os.open(path, os.O_RDONLY)
    m = mmap.mmap(fd, at, flags=mmap.MAP_PRIVATE, prot=mmap.PROT_READ | mmap.PROT_WRITE)
    whole = np.frombuffer(m, np.uint8)
    arr = {k: whole[o:o + n].view(np.asarray(rs[k]).dtype) for k, (o, n) in where.items()}
    d = Dfk(K=48)
    d.hint_file_range(whole, fd)
    whole[:] = 0xA5                                                       # (private: the file keeps the truth)
    d.count(arr["packed"], arr["base_off"], arr["read_len"], arr["pq_bytes"], arr["pq_off"], rs["bc"])
    assert d.stats()["n_solid"] == len(ref["solid"]) and np.array_equal(d.good_lens(), ref["good_len"])
    d.hint_file_range(None, -1)
    # without a descriptor: the memory is read (a shared mapping now, showing the file) and its pages are dropped behind the copy
    m2 = mmap.mmap(fd, at, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
    whole2 = np.frombuffer(m2, np.uint8)
    arr2 = {k: whole2[o:o + n].view(np.asarray(rs[k]).dtype) for k, (o, n) in where.items()}
    d.hint_file_range(whole2, -1)
    d.count(arr2["packed"], arr2["base_off"], arr2["read_len"], arr2["pq_bytes"], arr2["pq_off"], rs["bc"])
    assert d.stats()["n_solid"] == len(ref["solid"]) and np.array_equal(d.good_lens(), ref["good_len"])
    assert np.array_equal(arr2["read_len"], rs["read_len"])                 # (dropped pages come back from the file)
    d.close(); del arr, whole, arr2, whole2; m.close(); m2.close(); os.close(fd)


def test_dense_bases_and_barcode_index_are_expanded_on_the_device(oracle):
    """dfk_count with base_off = NULL (dense bases: the table is the running sum of ceil(len/4)) and dfk_count_bci (the barcode
    index of .bci instead of DF's expanded vector, DF.cc:447-452): the same dictionary as the explicit arrays give."""
    from superplus_amd.dfk import Dfk
    rs = util.make_set(93, 50000, 3000, ragged_frac=0.3)
    # ragged lengths: cut every seventh read short (dense layout again afterwards)
    from oracle import paths_oracle
    from superplus_amd import feudal
    reads, quals = paths_oracle.unpack_reads(rs)
    for i in range(0, len(reads), 7):
        reads[i] = reads[i][:61 + i % 30]; quals[i] = quals[i][:61 + i % 30]
    packed = np.concatenate([feudal.pack_bases(np.frombuffer(r, np.uint8)[None, :])[0] for r in reads])
    read_len = np.array([len(r) for r in reads], np.uint32)
    base_off = np.concatenate([[0], np.cumsum((read_len.astype(np.uint64) + 3) // 4)]).astype(np.uint64)
    pqs = [np.frombuffer(feudal.pq_encode(np.asarray(q, np.uint8)), np.uint8) for q in quals]
    pq_bytes = np.concatenate(pqs); pq_off = np.concatenate([[0], np.cumsum([len(x) for x in pqs])]).astype(np.uint64)
    # a barcode index with empty barcodes and a tail of reads beyond the last range
    n = len(read_len)
    cuts = sorted({0, 300, 300, 1200, 1210, 2500, n - 400})
    bci = np.array([0, 300, 300, 1200, 1210, 1210, 2500, n - 400], np.int64)
    bc = np.zeros(n, np.int32)
    for b in range(len(bci) - 1):
        bc[bci[b]:bci[b + 1]] = b
    ref = oracle.run(packed, base_off, read_len, pq_bytes, pq_off, bc, K=48, min_bc=1)
    for how in ("dense", "bci", "both"):
        d = Dfk(K=48, min_bc=1, keep_pre_adjacency=True)
        if how == "dense": d.count(packed, None, read_len, pq_bytes, pq_off, bc)
        elif how == "bci": d.count_bci(packed, base_off, read_len, pq_bytes, pq_off, bci)
        else: d.count_bci(packed, None, read_len, pq_bytes, pq_off, bci)
        util.check_parity(ref, d)
        d.close()
