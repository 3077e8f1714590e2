"""The multi-GPU (sharded) path on one GPU: every rank is a DistDfk context on cuda:0 and the
all-to-all exchanges are done by tensor slicing (superplus_amd.dist.run_inprocess).  The union of
the ranks' disjoint solid sets, the sum of their spectra and their goodLens must equal the oracle's
single-process result bit for bit."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _shards(rs, world, dev):
    n = rs["n_reads"]
    per = ((n // 2 + world - 1) // world) * 2                  # whole pairs
    out = []
    for r in range(world):
        a, b = min(n, r * per), min(n, (r + 1) * per)
        b0, b1 = int(rs["base_off"][a]), int(rs["base_off"][b])
        q0, q1 = int(rs["pq_off"][a]), int(rs["pq_off"][b])
        t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x).astype(dt)).to(dev)
        out.append((t(rs["packed"][b0:b1], np.uint8), t(rs["base_off"][a:b + 1] - rs["base_off"][a], np.int64),
                    t(rs["read_len"][a:b], np.int32), t(rs["pq_bytes"][q0:q1], np.uint8),
                    t(rs["pq_off"][a:b + 1] - rs["pq_off"][a], np.int64), t(rs["bc"][a:b], np.int32), a))
    return out


@pytest.mark.parametrize("world,K,ign,passes,pipelined", [(2, 48, 0, 0, False), (4, 48, 3000, 0, False), (8, 60, 0, 0, False),
                                                          (2, 40, 0, 0, False), (2, 48, 0, 4, False), (4, 48, 0, 2, False),
                                                          (2, 48, 0, 4, True), (4, 48, 0, 8, True), (8, 48, 0, 2, True)])
def test_sharded_equals_single(oracle, world, K, ign, passes, pipelined):
    """pipelined: the order of the real driver -- pass p+1 is cut and received (into the library's second receive
    buffer) before pass p is counted."""
    from superplus_amd.dist import DistDfk, run_inprocess
    rs = util.make_set(51 + world, 80000, 9000)
    ref = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"], K=K,
                     ign_bc_below=ign)
    dev = torch.device("cuda", 0)
    ranks = [DistDfk(K=K, device=0, ign_bc_below=ign, keep_pre_adjacency=True, passes=passes) for _ in range(world)]
    n_global = run_inprocess(ranks, _shards(rs, world, dev), pipelined=pipelined)
    assert n_global == ref["n_inst"]
    assert np.array_equal(np.concatenate([d.good_lens() for d in ranks]), ref["good_len"])
    for pre in (True, False):
        parts = [d.solid(pre_adjacency=pre) for d in ranks]
        allk = np.concatenate(parts)
        allk = allk[np.lexsort((allk["w1"], allk["w0"]))]
        util.assert_same_solid(allk, ref["solid_pre" if pre else "solid"], f"world={world} pre={pre}")
    assert min(len(d.solid()) for d in ranks) > 0               # every rank owns part of the set
    hist = np.zeros(len(ref["hist"]), np.int64)
    for d in ranks:
        h = d.spectrum(); hist[: len(h)] += h
    assert np.array_equal(hist, ref["hist"])
    assert sum(d.stats()["n_distinct"] for d in ranks) == ref["n_distinct"]


def test_sharded_long_reads(oracle):
    """Reads whose summaries overflow reach their owners through the scanning scatter's slice mode."""
    from superplus_amd.dist import DistDfk, run_inprocess
    rs = util.make_set(91, 200000, 3000, read_len=240)
    ref = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"], K=48)
    dev = torch.device("cuda", 0)
    ranks = [DistDfk(K=48, device=0, passes=2) for _ in range(2)]
    run_inprocess(ranks, _shards(rs, 2, dev), pipelined=True)
    allk = np.concatenate([d.solid() for d in ranks])
    util.assert_same_solid(allk[np.lexsort((allk["w1"], allk["w0"]))], ref["solid"], "sharded long reads")


def test_rccl_path_single_rank(oracle):
    """The real transport: torch.distributed with the nccl (= RCCL) backend, world size 1 on this GPU.
    Exercises TorchComm, the zero-copy views of library buffers and all_to_all_single with byte splits."""
    import os
    import torch.distributed as dist
    from superplus_amd.dist import DistDfk
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rs = util.make_set(71, 60000, 6000)
        ref = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"], K=48)
        d = DistDfk(K=48, device=0, passes=4)
        d.count_device(*_shards(rs, 1, dev)[0][:6])
        util.assert_same_solid(d.solid(), ref["solid"], "rccl world=1")
        assert np.array_equal(d.spectrum(), ref["hist"])
    finally:
        dist.destroy_process_group()


def _rank_main(rank, world, port, shard, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from superplus_amd.dist import DistDfk
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        t = [torch.from_numpy(x).to(dev) for x in shard[:6]]
        d = DistDfk(K=48, device=0, passes=2, hbm_budget_bytes=8 << 30)
        d.count_device(*t, read_id0=shard[6])
        q.put((rank, "ok", d.solid(), np.asarray(d.spectrum()), d.good_lens(), d.stats()["n_inst_global"]))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc(), None, None, None, 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_one_process_per_rank(oracle, world):
    """DistDfk.count_device as bench.py runs it -- one process per rank, a torch.distributed group, TorchComm's
    point-to-point rounds, the pipelined passes -- with the ranks sharing this GPU and gloo as the transport
    (RCCL does not put two ranks on one device; TorchComm stages device buffers through the host for gloo)."""
    import torch.multiprocessing as mp
    rs = util.make_set(77, 300000, 20000)
    ref = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"], K=48)
    shards = [tuple(x.cpu().numpy() if torch.is_tensor(x) else x for x in sh) for sh in _shards(rs, world, "cpu")]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + int(np.random.default_rng().integers(0, 200))
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, shards[r], q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    finally:
        for p in procs:                                  # a rank that is still there (hung in a collective) must not outlive the test
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    allk = np.concatenate([r[2] for r in res])
    util.assert_same_solid(allk[np.lexsort((allk["w1"], allk["w0"]))], ref["solid"], "one process per rank")
    n = max(len(r[3]) for r in res)
    hist = sum(np.pad(r[3], (0, n - len(r[3]))) for r in res)
    assert np.array_equal(hist, ref["hist"])
    assert np.array_equal(np.concatenate([r[4] for r in res]), ref["good_len"])
    assert all(r[5] == ref["n_inst"] for r in res)
