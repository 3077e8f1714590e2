"""world_size-2 gloo test of the multi-GPU exchange plumbing (runs on the CPU).

The kernels need a GPU, but the routing logic around them does not: this checks that
exchange() delivers every rank's per-destination slices to the right peer in source-rank
order, and that the answer round trip returns bytes in the order the queries were sent --
the two properties the sharded recomputeAdjacencies relies on."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q, piece):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from superplus_amd.dist import TorchComm, exchange
        comm = TorchComm()
        if piece:
            comm.PIECE = piece                              # slices of up to 49 units of 32 bytes: several rounds each
        rng = np.random.default_rng(100 + rank)
        # records: 32-byte units; unit r->d carries (src, dst, serial) so the receiver can check routing
        counts = [int(rng.integers(0, 50)) for _ in range(world)]
        if rank == 1:
            counts[0] = 0                                   # an empty slice must work too
        units = []
        for d in range(world):
            for s in range(counts[d]):
                u = np.zeros(32, np.uint8); u[0], u[1], u[2], u[3] = rank, d, s & 255, s >> 8
                units.append(u)
        send = torch.from_numpy(np.concatenate(units) if units else np.zeros(0, np.uint8))
        recv, rcounts = exchange(send, counts, 32, comm)
        got = recv.numpy().reshape(-1, 32)
        pos = 0
        for src in range(world):
            for s in range(rcounts[src]):
                assert (got[pos][0], got[pos][1], got[pos][2] | (got[pos][3] << 8)) == (src, rank, s)
                pos += 1
        assert pos == len(got)
        # query/answer round trip: answer = f(key) computed by the owner, must come back in query order
        keys = torch.from_numpy(rng.integers(0, 255, 16 * sum(counts), dtype=np.uint8))
        rkeys, rc = exchange(keys, counts, 16, comm)
        answers = (rkeys.reshape(-1, 16).sum(dim=1) % 251).to(torch.uint8) + rank  # owner-specific
        back, _ = exchange(answers, rc, 1, comm)
        owner = np.repeat(np.arange(world), counts)
        expect = (keys.reshape(-1, 16).sum(dim=1) % 251).to(torch.uint8).numpy() + owner.astype(np.uint8)
        assert np.array_equal(back.numpy(), expect)
        # the pipelined form the driver uses: two exchanges begun back to back, ended in order, while "work" happens
        from superplus_amd.dist import exchange_begin, exchange_end
        t1 = exchange_begin(send, counts, 32, comm)
        t2 = exchange_begin(keys, counts, 16, comm)
        r1, c1 = exchange_end(t1)
        r2, c2 = exchange_end(t2)
        assert np.array_equal(r1.numpy(), recv.numpy()) and c1 == rcounts
        assert np.array_equal(r2.numpy(), rkeys.numpy()) and c2 == rc
        total = comm.all_reduce_sum(sum(counts), "cpu")
        q.put((rank, "ok", total))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc(), 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,piece", [(2, 0), (4, 0), (2, 96), (3, 40)])
def test_exchange_routing_gloo(world, piece):
    """piece: the largest point-to-point message in bytes (0 = the default, 1 GiB); the small values split every
    slice over several rounds, with a different number of rounds for every pair of ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + int(np.random.default_rng().integers(0, 2000))
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, piece)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    assert len({r[2] for r in res}) == 1


def _single(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from superplus_amd.dist import TorchComm, exchange
        comm = TorchComm()
        send = torch.arange(96, dtype=torch.uint8)
        recv, rc = exchange(send, [3], 32, comm)                   # a rank alone keeps its slice: a copy, no message
        assert rc == [3] and torch.equal(recv, send)
        out = torch.empty(64, dtype=torch.uint8)
        for bad in (([32], [96]), ([64], [64])):                   # splits that do not cover the buffers / do not agree
            try:
                comm.all_to_all_single(out, send, bad[0], bad[1])
                q.put("no error for %r" % (bad,)); return
            except ValueError:
                pass
        q.put("ok")
    except Exception:  # pragma: no cover
        import traceback
        q.put("fail: " + traceback.format_exc())
    finally:
        dist.destroy_process_group()


def test_single_rank_and_bad_splits():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single, args=(29400 + int(np.random.default_rng().integers(0, 90)), q))
    p.start()
    res = q.get(timeout=120)
    p.join(timeout=60)
    assert res == "ok", res


def _failing_rank(rank, world, port, q, fail_in):
    """DistDfk.count_device with the library calls replaced by CPU stand-ins; rank 1 fails in `fail_in`."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from superplus_amd import dist as D
        from superplus_amd.dfk import DfkError

        class Stub(D.DistDfk):
            def __init__(self):                         # no context: every library call is overridden
                self.comm = None; self._n_inst_global = 0; self._ctx = None

            def _maybe(self, what):
                if rank == 1 and what == fail_in:
                    raise DfkError(-4, f"injected failure in {what}")

            def begin(self, *a, **k): self._maybe("begin"); return 1000
            def plan(self, world, n): self._maybe("plan"); return 2            # four passes: the pipeline reaches its steady state
            def partition(self, world, n, l, p): self._maybe("partition%d" % p); return torch.zeros(32 * 3 * world, dtype=torch.uint8), [3] * world
            def partition_begin(self, world, n, l, p, defer): return self.partition(world, n, l, p)
            def partition_end(self, p): self._maybe("partend%d" % p)
            def recv_buffer(self, n): self._maybe("recv"); return torch.zeros(32 * n, dtype=torch.uint8)
            def count_records(self, recv, p): self._maybe("count%d" % p)
            def adj_queries(self, world): self._maybe("adjq"); return torch.zeros(16 * 2 * world, dtype=torch.uint8), [2] * world
            def adj_answer(self, keys): self._maybe("adja"); return torch.ones(keys.numel() // 16, dtype=torch.uint8)
            def adj_apply(self, present): pass
            def close(self): pass

        t = torch.zeros(4, dtype=torch.uint8)
        try:
            Stub().count_device(t, t, t, t, t, t)
            q.put((rank, "completed"))
        except DfkError as e:
            q.put((rank, "DfkError %d" % e.code))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail_in", ["none", "begin", "plan", "partition0", "recv", "count0", "partition1", "count1", "partition2", "partend2",
                                     "partition3", "count2", "count3", "adjq", "adja"])
def test_one_rank_failing_stops_every_rank(fail_in):
    """A library call that fails on one rank only (its own data, its own HBM budget) must end the run on EVERY rank
    with an error, not leave the others inside the next collective: the driver all-reduces a status
    word after each local phase."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + int(np.random.default_rng().integers(0, 2000))
    procs = [ctx.Process(target=_failing_rank, args=(r, 2, port, q, fail_in)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = dict(q.get(timeout=90) for _ in range(2))
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    if fail_in == "none":
        assert res == {0: "completed", 1: "completed"}, res
    else:
        assert res == {0: "DfkError -4", 1: "DfkError -4"}, res


def test_cpp_exchange_schedule(tmp_path):
    """The C++ host's all-to-all schedule (superplus_amd/csrc/dfk_exchange.h: pieces in rounds, rank r to r+d while it
    receives from r-d) over a loopback transport, every rank a thread: tests/cpp/test_exchange.cc."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_exchange")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-Wall", "-o", exe, os.path.join(root, "tests", "cpp", "test_exchange.cc")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "FAILED" not in out.stdout and out.stdout.count(": ok") >= 20, out.stdout + out.stderr
