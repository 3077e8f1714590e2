"""BASELINE-scale properties of the HIP path that no oracle run can check: the dictionary must not depend on how the
minimizer buckets are cut into passes, nor on which path a hot bucket takes -- including beyond 2^31 records in one
pass block (68.7 GB), where a 32-bit quantity anywhere in the kernels shows."""
import pytest
import torch

from superplus_amd import synth
from superplus_amd.dfk import Dfk

pytestmark = pytest.mark.gpu


def _counts(shard, passes, K=48):
    d = Dfk(K=K, device=0, passes=passes)
    d.count_device(*shard)
    st, dg = d.stats(), d.digest()
    d.close()
    torch.cuda.empty_cache()
    return (st["n_inst"], st["n_distinct"], st["n_solid"], dg), st


def _reads(genome_mb, copies, seed):
    dev = torch.device("cuda:0")
    G = int(genome_mb * 1e6)
    genome = synth.make_genome(G, seed, device=dev, family_copies=copies, low_complexity_frac=0.01 if copies else 0.0)
    rs = synth.make_reads(genome, int(30 * G / 200), seed + 17, ragged_frac=0.25 if copies else 0.0)
    del genome
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    return (rs.packed, rs.base_off, rs.read_len, rs.pq_bytes, rs.pq_off, rs.bc)


def _need_hbm(gb):
    import gc, time
    gc.collect(); torch.cuda.empty_cache()
    for _ in range(60):                                                    # (released memory is wiped by the driver before it is free again: wait, do not skip)
        free, _ = torch.cuda.mem_get_info(0)
        if free >= gb * 1e9: break
        time.sleep(1.0); torch.cuda.empty_cache()
    assert free >= gb * 1e9, f"needs {gb} GB of free HBM ({free / 1e9:.0f} free)"


def test_one_pass_of_more_than_2_pow_31_records_equals_eight_passes():
    """2.1 Gb genome at 30x: 2.3e9 super-k-mer records.  Counted as ONE pass its record block is 74 GB and record
    indices pass 2^31 (k_count once widened a readfirstlane'd low word as a signed int: such a pass read 64 GiB below
    its block)."""
    _need_hbm(250)
    shard = _reads(2100, 0, 4101)
    one, st1 = _counts(shard, 1)
    assert st1["n_passes"] == 1 and st1["n_records"] > 1 << 31
    eight, st8 = _counts(shard, 8)
    assert st8["n_passes"] == 8
    assert one == eight


def test_hot_buckets_beyond_2_pow_31_instances_in_one_pass():
    """A 620 Mb genome half made of one diverged 300-bp family: in one pass 55 k hot buckets hold 3.6e9 instances, which
    the second-level partition (k_hot_split) writes out as 114 GB of one-k-mer records; in four passes a quarter each."""
    _need_hbm(250)
    shard = _reads(620, 1_033_333, 20250)
    one, st1 = _counts(shard, 1)
    four, st4 = _counts(shard, 4)
    assert st1["n_passes"] == 1 and st4["n_passes"] == 4
    assert one == four
    again, _ = _counts(shard, 4)                      # and from run to run
    assert again == four


@pytest.mark.parametrize("K", [40, 60])
def test_other_k_with_hot_buckets_one_pass_equals_four(K):
    """The same property at the other two k-mer widths (K = 60 keys are four words wide: its own table layout, its own
    instantiation of every kernel), on a genome a third of which is one repeat family: the hot-bucket paths (second-level
    partition, one-k-mer records, HBM tables) run at scale for each K."""
    _need_hbm(200)
    shard = _reads(310, 344_444, 777 + K)
    one, st1 = _counts(shard, 1, K)
    four, st4 = _counts(shard, 4, K)
    assert st1["n_passes"] == 1 and st4["n_passes"] == 4 and st1["n_overflow_items"] > 1000
    assert one == four
