"""BASELINE configs[1] at full size (900 M pairs over a 3.1 Gb genome), checked through properties that do not need
the oracle to run at that size:

  * the dictionary does not depend on how the work was cut: the passes the planner chooses, twice as many forced
    passes, and the sharded pipeline (one rank that owns every bucket, the records going through the exchange
    buffers) give the same digest, distinct count and spectrum;
  * the spectrum is the dictionary's: its bins add up to the solid count, nothing below MIN_FREQ;
  * the instance count is what goodLens imply (sum of max(0, goodLen - K + 1)).

The digest (dfk_solid_digest) is tied to the oracle's entries at small sizes by tests/util.check_parity, which every
parity test goes through."""
import numpy as np
import pytest
import torch

from superplus_amd import synth

pytestmark = pytest.mark.gpu

G, PAIRS, K = 3_100_000_000, 900_000_000, 48


def _summary(d):
    st = d.stats()
    return {"n_inst": st["n_inst"], "n_distinct": st["n_distinct"], "n_solid": st["n_solid"],
            "digest": d.digest(), "spectrum": np.asarray(d.spectrum()).tolist()}


def test_configs1_full_size_properties():
    from superplus_amd.dfk import Dfk
    from superplus_amd.dist import DistDfk, run_inprocess
    import gc, time
    dev = torch.device("cuda", 0)
    gc.collect(); torch.cuda.empty_cache()
    for _ in range(60):                                                    # (memory an earlier test released is wiped by the driver before it is free again)
        free, _ = torch.cuda.mem_get_info(dev)
        if free >= 240e9: break
        time.sleep(1.0); torch.cuda.empty_cache()
    assert free >= 240e9, "needs a whole MI355X (%.0f GB free)" % (free / 1e9)
    genome = synth.make_genome(G, 20261004, device=dev)
    rs = synth.make_reads(genome, PAIRS, 20261021)
    del genome
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    shard = (rs.packed, rs.base_off, rs.read_len, rs.pq_bytes, rs.pq_off, rs.bc)

    d = Dfk(K=K, device=0)
    d.count_device(*shard)
    a = _summary(d)
    passes = d.stats()["n_passes"]
    spec = np.asarray(a["spectrum"], dtype=np.int64)
    assert a["n_solid"] > 2_000_000_000                                   # ~the genome's k-mers
    assert int(spec.sum()) == a["n_solid"] and not spec[:3].any()         # MIN_FREQ = 3
    gl = d.good_lens().astype(np.int64)
    assert int(np.maximum(gl - (K - 1), 0).sum()) == a["n_inst"]
    del gl, d
    torch.cuda.empty_cache()

    d = Dfk(K=K, device=0, passes=2 * passes)
    d.count_device(*shard)
    assert d.stats()["n_passes"] == 2 * passes
    b = _summary(d)
    assert b == a, "twice as many bucket-range passes"
    del d
    torch.cuda.empty_cache()

    # the sharded pipeline against the single-GPU one on the share of the reads one of four ranks would hold
    # (it keeps a second copy of its dictionary while merging the passes' parts: sized for a rank's share,
    # not for the whole set) -- 14 GB of records per pass through the exchange buffers
    n = PAIRS // 2                                                         # reads = a quarter of the pairs
    nb, nq = int(rs.base_off[n].item()), int(rs.pq_off[n].item())
    part = (rs.packed[:nb], rs.base_off[: n + 1], rs.read_len[:n], rs.pq_bytes[:nq], rs.pq_off[: n + 1], rs.bc[:n])
    d = Dfk(K=K, device=0)
    d.count_device(*part)
    a = _summary(d)
    del d
    torch.cuda.empty_cache()
    s = DistDfk(K=K, device=0)
    run_inprocess([s], [part + (0,)], pipelined=True)
    assert s.stats()["n_passes"] > 1
    c = _summary(s)
    assert c == a, "sharded pipeline, one rank"
