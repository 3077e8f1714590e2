"""Multi-GPU driver: one process per GPU, the exchanges over torch.distributed (backend "nccl"
is RCCL on ROCm, over xGMI inside a node).

The reference's only exchange is MapReduceEngine's thread all-to-all (MapReduceEngine.h:345-388).
Here: reads are sharded by contiguous pair ranges; every rank cuts its reads into super-k-mer
records and sends each to the rank that owns its minimizer bucket (ONE all-to-all of 32-byte
records); counting is then local; recomputeAdjacencies needs neighbours that live on other
ranks, which costs one small all-to-all of 16-byte keys and one of 1-byte answers.

`Comm` hides the transport so the same driver runs (a) under torch.distributed and (b) with all
ranks inside one process, which is how the path is tested on a single GPU and how the
exchange plumbing is tested on the CPU with gloo.
"""
import ctypes as C

import os

import numpy as np
import torch

from . import dfk as _dfk
from .dfk import Dfk, _check, lib


def exchange_begin(send: torch.Tensor, send_counts, unit: int, comm, recv_alloc=None, agree=None):
    """First half of exchange(): the counts are exchanged (small, waited for), the receive buffer is made and the
    all-to-all of the payload is started.  Returns a token for exchange_end().  With a transport that cannot run
    the collective in the background the whole exchange happens here.
    agree(err): called with the DfkError recv_alloc raised (or None) before the payload moves; it is expected to
    make every rank raise if any rank failed (the receive buffer comes out of each rank's own HBM budget)."""
    sc = torch.tensor(list(send_counts), dtype=torch.int64, device=send.device)
    rc = torch.empty_like(sc)
    comm.all_to_all_single(rc, sc, None, None)
    rcl = [int(x) for x in rc.tolist()]
    err, recv = None, None
    try:
        recv = recv_alloc(sum(rcl)) if recv_alloc else torch.empty(sum(rcl) * unit, dtype=torch.uint8, device=send.device)
    except _dfk.DfkError as e:
        err = e
    if agree is not None:
        agree(err)
    elif err is not None:
        raise err
    if send.is_cuda:
        torch.cuda.synchronize(send.device)       # the library filled `send` on its own stream
    start = getattr(comm, "all_to_all_single_async", None)
    work = None
    if start is not None:
        work = start(recv, send, [x * unit for x in rcl], [int(x) * unit for x in send_counts])
    else:
        comm.all_to_all_single(recv, send, [x * unit for x in rcl], [int(x) * unit for x in send_counts])
    return recv, rcl, work, send                   # `send` is kept alive until the collective has finished


def exchange_end(token):
    recv, rcl, work, send = token
    if work is not None:
        work.wait()
    if recv.is_cuda:
        torch.cuda.synchronize(recv.device)        # the library reads `recv` on its own stream
    return recv, rcl


def exchange(send: torch.Tensor, send_counts, unit: int, comm, recv_alloc=None):
    """All-to-all of variable-size slices.  `send` is a flat byte tensor holding, rank after rank,
    send_counts[r] units of `unit` bytes for rank r.  Returns (recv bytes, recv_counts).
    recv_alloc(n_units) -> byte tensor lets the caller supply the receive buffer (the record
    exchange receives into the library's own HBM budget)."""
    world = comm.world
    sc = torch.tensor(list(send_counts), dtype=torch.int64, device=send.device)
    rc = torch.empty_like(sc)
    comm.all_to_all_single(rc, sc, None, None)
    rcl = [int(x) for x in rc.tolist()]
    recv = recv_alloc(sum(rcl)) if recv_alloc else torch.empty(sum(rcl) * unit, dtype=torch.uint8, device=send.device)
    if send.is_cuda:
        torch.cuda.synchronize(send.device)       # the library filled `send` on its own stream
    comm.all_to_all_single(recv, send, [x * unit for x in rcl], [int(x) * unit for x in send_counts])
    if send.is_cuda:
        torch.cuda.synchronize(send.device)       # ... and reads `recv` on it: the collective must have finished
    return recv, rcl


class TorchComm:
    """torch.distributed transport (RCCL on GPUs, gloo in the CPU tests).

    The payload never goes through the backend's all-to-all in one piece.  RCCL 2.26 (ROCm 7.0) delivers only the
    first half of a send/recv of 2.5 GB or more and reports success (tools/check_rccl_large.py: rank to itself,
    as bytes or as int64, waited for or not), and a pass of BASELINE configs[2] moves 3-7 GB per pair of ranks at
    2 and 4 ranks.  So: the slice a rank keeps is a device copy, and every other slice travels as point-to-point
    messages of at most PIECE bytes, one batch (= one RCCL group) per round."""

    PIECE = int(os.environ.get("DFK_A2A_PIECE_BYTES", str(1 << 30)))

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # gloo moves host memory only: device buffers are staged through the host (the two-process GPU test; a
        # real run uses nccl)
        self.staged = dist.get_backend(group) == "gloo"

    def _start(self, out, inp, out_split, in_split):
        """Start the exchange; returns the requests still in flight."""
        if self.staged and inp.is_cuda:
            h_out = torch.empty(out.shape, dtype=out.dtype)
            for r in self._start(h_out, inp.cpu(), out_split, in_split):
                r.wait()
            out.copy_(h_out)
            return []
        world, rank, dist = self.world, self.rank, self.dist
        if out_split is None:
            out_split = [out.numel() // world] * world
        if in_split is None:
            in_split = [inp.numel() // world] * world
        if sum(out_split) != out.numel() or sum(in_split) != inp.numel():
            raise ValueError("all_to_all: the splits do not cover the buffers")
        out_off = np.concatenate([[0], np.cumsum(out_split)]).astype(np.int64)
        in_off = np.concatenate([[0], np.cumsum(in_split)]).astype(np.int64)
        if out_split[rank] != in_split[rank]:
            raise ValueError("all_to_all: this rank sends itself %d elements and expects %d" % (in_split[rank], out_split[rank]))
        if in_split[rank]:
            out[int(out_off[rank]) : int(out_off[rank + 1])].copy_(inp[int(in_off[rank]) : int(in_off[rank + 1])])
        piece = max(1, self.PIECE // max(1, inp.element_size()))
        largest = max([n for r, n in enumerate(out_split) if r != rank] + [n for r, n in enumerate(in_split) if r != rank] + [0])
        reqs = []
        for j in range((largest + piece - 1) // piece):
            ops = []
            for d in range(1, world):           # rank r sends to r+d while it receives from r-d: every pair posts in the same order
                to, frm = (rank + d) % world, (rank - d) % world
                lo, hi = j * piece, (j + 1) * piece
                if lo < in_split[to]:
                    ops.append(dist.P2POp(dist.isend, inp[int(in_off[to]) + lo : int(in_off[to]) + min(hi, in_split[to])],
                                          self._peer(to), self.group))
                if lo < out_split[frm]:
                    ops.append(dist.P2POp(dist.irecv, out[int(out_off[frm]) + lo : int(out_off[frm]) + min(hi, out_split[frm])],
                                          self._peer(frm), self.group))
            if ops:
                reqs += dist.batch_isend_irecv(ops)
        return reqs

    def _peer(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def all_to_all_single(self, out, inp, out_split, in_split):
        for r in self._start(out, inp, out_split, in_split):
            r.wait()

    def all_to_all_single_async(self, out, inp, out_split, in_split):
        """The same exchange, left running: returns a handle whose wait() ends it."""
        reqs = self._start(out, inp, out_split, in_split)

        class _Pending:
            def wait(self_inner):
                for r in reqs:
                    r.wait()
        return _Pending()

    def all_reduce_sum(self, value: int, device):
        t = torch.tensor([value], dtype=torch.int64, device="cpu" if self.staged else device)
        self.dist.all_reduce(t, group=self.group)
        return int(t.item())

    def all_reduce_max(self, value: int, device):
        t = torch.tensor([value], dtype=torch.int64, device="cpu" if self.staged else device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())


class ReplicaComm:
    """One rank of `world`, alone on its GPU: every peer is taken to hold the same reads as this rank, so
    each of them sends this rank exactly what this rank sends to itself.  For rehearsing one rank's memory
    plan and compute time at full scale on a single GPU (bench.py --emulate-world): the owned k-mers get
    their full coverage, the volumes received are those of a real run; the transfers themselves cost a
    device copy.  Answers to this rank's neighbour queries to other ranks are made up (all present)."""
    rehearsal = True

    def __init__(self, world):
        self.rank, self.world = 0, world

    def all_to_all_single(self, out, inp, out_split, in_split):
        if in_split is None:                       # the counts: everybody sends what we send to ourselves
            out.copy_(inp[:1].expand_as(out))
            return
        mine = inp[: in_split[0]]
        at = 0
        for n in out_split:
            out[at : at + n].copy_(mine[:n])
            at += n

    def all_reduce_sum(self, value, device):
        return value * self.world

    def all_reduce_max(self, value, device):
        return value


def _view(ptr, nbytes, device):
    """A uint8 torch view of library-owned device memory (no copy)."""
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device=device)
    class _Holder:  # __cuda_array_interface__ carrier
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    return torch.as_tensor(h, device=device)


class DistDfk(Dfk):
    """Dfk whose count_device() runs the sharded pipeline.  After it returns, solid()/spectrum()/
    good_lens() give THIS rank's share: solid sets of different ranks are disjoint, spectra add."""

    def __init__(self, comm=None, **kw):
        super().__init__(**kw)
        self.comm = comm
        self._n_inst_global = 0

    # ---- phases (each is local; the driver below puts the collectives between them) ----
    def begin(self, packed, base_off, read_len, pq_bytes, pq_off, bc, read_id0=0):
        n = C.c_uint64()
        dp = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
        _check(lib().dfk_shard_begin(self._ctx, dp(packed), C.c_uint64(packed.numel()), dp(base_off), dp(read_len),
                                     dp(pq_bytes), C.c_uint64(pq_bytes.numel()), dp(pq_off), dp(bc),
                                     C.c_uint64(read_len.numel()), C.c_int64(read_id0), C.byref(n)))
        self._device = packed.device
        return n.value

    def plan(self, world, n_inst_global):
        l = C.c_uint32()
        _check(lib().dfk_shard_plan(self._ctx, C.c_uint32(world), C.c_uint64(n_inst_global), C.byref(l)))
        return l.value

    def partition(self, world, n_inst_global, log2_passes=0, pass_=0):
        ptr = C.c_void_p(); counts = (C.c_uint64 * world)()
        _check(lib().dfk_shard_partition(self._ctx, C.c_uint32(world), C.c_uint64(n_inst_global), C.c_uint32(log2_passes),
                                         C.c_uint32(pass_), C.byref(ptr), counts))
        counts = list(counts)
        return _view(ptr.value, 32 * sum(counts), self._device), counts

    def partition_begin(self, world, n_inst_global, log2_passes, pass_, defer):
        """Room and send counts of pass `pass_` at once; its kernels start now, or (defer) behind the k_count of the
        count_records call that follows, on the library's second stream.  The buffer may be sent after partition_end."""
        ptr = C.c_void_p(); counts = (C.c_uint64 * world)()
        _check(lib().dfk_shard_partition_begin(self._ctx, C.c_uint32(world), C.c_uint64(n_inst_global), C.c_uint32(log2_passes),
                                               C.c_uint32(pass_), C.c_int(1 if defer else 0), C.byref(ptr), counts))
        counts = list(counts)
        return _view(ptr.value, 32 * sum(counts), self._device), counts

    def partition_end(self, pass_):
        _check(lib().dfk_shard_partition_end(self._ctx, C.c_uint32(pass_)))

    def recv_buffer(self, n_records):
        ptr = C.c_void_p()
        _check(lib().dfk_shard_recv_buffer(self._ctx, C.c_uint64(n_records), C.byref(ptr)))
        return _view(ptr.value, 32 * n_records, self._device)

    def count_records(self, recv, pass_=0):
        _check(lib().dfk_shard_count(self._ctx, C.c_void_p(recv.data_ptr() if recv.numel() else 0), C.c_uint64(recv.numel() // 32),
                                     C.c_uint32(pass_)))

    def adj_queries(self, world):
        ptr = C.c_void_p(); counts = (C.c_uint64 * world)()
        _check(lib().dfk_shard_adj_queries(self._ctx, C.byref(ptr), counts))
        counts = list(counts)
        return _view(ptr.value, 16 * sum(counts), self._device), counts

    def adj_answer(self, keys):
        n = keys.numel() // 16
        present = torch.empty(n, dtype=torch.uint8, device=self._device)
        _check(lib().dfk_shard_adj_answer(self._ctx, C.c_void_p(keys.data_ptr() if n else 0), C.c_uint64(n),
                                          C.c_void_p(present.data_ptr() if n else 0)))
        return present

    def adj_apply(self, present):
        _check(lib().dfk_shard_adj_apply(self._ctx, C.c_void_p(present.data_ptr() if present.numel() else 0),
                                         C.c_uint64(present.numel())))

    # ---- the sharded createDict ----
    def count_device(self, packed, base_off, read_len, pq_bytes, pq_off, bc, read_id0=0):
        """Runs the sharded pipeline.  Afterwards self.timing holds this rank's host-clock milliseconds per phase of
        the call (every library call ends with the device idle, so the host clock is the device's):
        trim, plan (the counting scan), partition (sweeps), exchange_wait (time this rank sat waiting for records:
        what the transfers cost beyond what hides under the count), count (regroup + count), adjacency (queries,
        both exchanges, answers), total; and the bytes it sent to other ranks."""
        import time
        T = {k: 0.0 for k in ("trim", "plan", "partition", "exchange_wait", "count", "adjacency", "total")}
        sent = 0
        t_all = time.perf_counter()

        comm = self.comm or TorchComm()
        world = comm.world
        pending = []                                   # a local failure waits here for the next agreement point

        def timed(key, f, *a):
            t = time.perf_counter()
            r = None
            try:
                if not pending:
                    r = f(*a)
            except _dfk.DfkError as e:                 # the library's own errors only: they are local to this rank
                pending.append(e)
            T[key] += 1e3 * (time.perf_counter() - t)
            return r

        def agree(what):
            """A library call can fail on one rank only (its own data, its own HBM budget).  The others would then
            sit in the next collective until the backend times out, holding their GPUs.  So after every local phase
            the ranks all-reduce a status word and every rank raises when any of them failed -- before the next
            exchange is entered."""
            worst = comm.all_reduce_max(-pending[0].code if pending else 0, packed.device)
            if pending:
                raise pending[0]
            if worst:
                raise _dfk.DfkError(-worst, f"another rank failed in {what}; this rank stops with it")

        n_local = timed("trim", self.begin, packed, base_off, read_len, pq_bytes, pq_off, bc, read_id0)
        agree("dfk_shard_begin")
        n_global = comm.all_reduce_sum(n_local, packed.device)
        if n_global == 0:
            raise _dfk.DfkError(-7, "Looks like your input data have almost no good bases.")
        self._n_inst_global = n_global
        mine = timed("plan", self.plan, world, n_global)
        agree("dfk_shard_plan")
        log2_passes = comm.all_reduce_max(mine, packed.device)   # every rank runs the same passes
        # The k-mer shuffle, one bucket range at a time -- and two ranges ahead: while pass p is counted, the records of
        # pass p+1 are on their way (the library keeps a second receive buffer for them) and those of pass p+2 are being
        # cut, by a sweep the library starts behind the count's k_count on its second stream (a second send buffer).
        n_pass = 1 << log2_passes
        serial = bool(os.environ.get("DFK_SHARD_SERIAL"))          # debugging aid: nothing under the count
        def cut(p):
            r = timed("partition", self.partition, world, n_global, log2_passes, p)
            agree("dfk_shard_partition")               # (also covers the count of the pass before: see below)
            return r

        send, counts = cut(0)
        sent += 32 * (sum(counts) - counts[comm.rank])
        def agree_recv(err):
            if err is not None:
                pending.append(err)
            agree("dfk_shard_recv_buffer")

        def wait_for(token):                           # (never skipped: a transfer that was started is waited for)
            t = time.perf_counter()
            r = exchange_end(token)
            T["exchange_wait"] += 1e3 * (time.perf_counter() - t)
            return r

        def start(send, counts):                       # (contains collectives: never skipped, never swallowed)
            t = time.perf_counter()
            token = exchange_begin(send, counts, 32, comm, self.recv_buffer, agree_recv)
            T["exchange_wait"] += 1e3 * (time.perf_counter() - t)
            return token

        recv, _ = wait_for(start(send, counts))
        ahead = cut(1) if n_pass > 1 else None         # (nothing to hide under yet)
        for p in range(n_pass):
            token = None
            if ahead is not None:                      # pass p+1: partitioned, checked and agreed on; send it off
                send, counts = ahead
                sent += 32 * (sum(counts) - counts[comm.rank])
                token = start(send, counts)
                if serial:
                    token = (*wait_for(token), None, None)
            ahead = None
            if p + 2 < n_pass:
                if serial:
                    ahead = cut(p + 2)
                else:
                    ahead = timed("partition", self.partition_begin, world, n_global, log2_passes, p + 2, True)
            # a failure here is agreed on below -- after the exchange that is already in flight has been waited for,
            # so that no rank is left inside a transfer
            timed("count", self.count_records, recv, p)
            del recv
            if p + 2 < n_pass and not serial:
                timed("partition", self.partition_end, p + 2)
            if token is not None:
                recv, _ = wait_for(token)
                del token
            if p + 2 < n_pass and not serial:
                agree("dfk_shard_count / dfk_shard_partition")
        del send
        t_adj = time.perf_counter()
        q = timed("adjacency", self.adj_queries, world)
        agree("dfk_shard_count / dfk_shard_adj_queries")
        keys, kcounts = q
        sent += 17 * (sum(kcounts) - kcounts[comm.rank])
        rkeys, rcounts = exchange(keys, kcounts, 16, comm)                 # neighbour queries
        answers = timed("adjacency", self.adj_answer, rkeys)
        agree("dfk_shard_adj_answer")
        if getattr(comm, "rehearsal", False):                              # no peers to answer: see ReplicaComm
            back = torch.ones(sum(kcounts), dtype=torch.uint8, device=packed.device)
        else:
            back, _ = exchange(answers, rcounts, 1, comm)                  # answers return in query order
        self.adj_apply(back)
        T["adjacency"] = 1e3 * (time.perf_counter() - t_adj)
        T["total"] = 1e3 * (time.perf_counter() - t_all)
        T["bytes_sent_to_peers"] = sent
        T["n_passes"] = n_pass
        self.timing = T

    def stats(self):
        s = super().stats()
        s["n_inst_global"] = self._n_inst_global
        return s


class _LocalComm:
    """All ranks in one process: collectives are performed by run_inprocess() between phases."""
    def __init__(self, rank, world):
        self.rank, self.world = rank, world


def _a2a(bufs, counts, unit):
    """bufs[r] = flat bytes of rank r grouped by destination; counts[r][d] units.  -> per-destination concatenation."""
    world = len(bufs)
    offs = [np.concatenate([[0], np.cumsum(c)]) * unit for c in counts]
    out, rcounts = [], []
    for d in range(world):
        parts = [bufs[r][int(offs[r][d]):int(offs[r][d + 1])] for r in range(world)]
        out.append(torch.cat(parts) if parts else bufs[0][:0])
        rcounts.append([counts[r][d] for r in range(world)])
    return out, rcounts


def run_inprocess(ranks, shards, pipelined=False):
    """Drive `ranks` (DistDfk objects, possibly all on one GPU) through the sharded pipeline with the
    exchanges done by tensor slicing.  shards[r] = (packed, base_off, read_len, pq_bytes, pq_off, bc, read_id0).
    pipelined: the order DistDfk.count_device uses (pass p+1 is cut and received before pass p is counted,
    into buffers the library hands out)."""
    world = len(ranks)
    n_local = [ranks[r].begin(*shards[r]) for r in range(world)]
    n_global = sum(n_local)
    log2_passes = max(ranks[r].plan(world, n_global) for r in range(world))

    def deliver(sends):
        recv, _ = _a2a([s[0] for s in sends], [s[1] for s in sends], 32)
        out = []
        for r in range(world):                  # receive into the library's own buffers, as the real driver does
            buf = ranks[r].recv_buffer(recv[r].numel() // 32)
            buf.copy_(recv[r])
            out.append(buf)
        return out

    if pipelined:
        # the order DistDfk.count_device uses: pass p+1 travels and pass p+2 is cut (behind the k_count) while p is counted
        n_pass = 1 << log2_passes
        recv = deliver([ranks[r].partition(world, n_global, log2_passes, 0) for r in range(world)])
        ahead = [ranks[r].partition(world, n_global, log2_passes, 1) for r in range(world)] if n_pass > 1 else None
        for p in range(n_pass):
            nxt = deliver(ahead) if ahead is not None else None
            ahead = [ranks[r].partition_begin(world, n_global, log2_passes, p + 2, True) for r in range(world)] if p + 2 < n_pass else None
            for r in range(world):
                ranks[r].count_records(recv[r], p)
                ranks[r]._n_inst_global = n_global
                if p + 2 < n_pass:
                    ranks[r].partition_end(p + 2)
            recv = nxt
        return _inprocess_adjacency(ranks, world, n_global)

    def shuffle(p):
        sends = [ranks[r].partition(world, n_global, log2_passes, p) for r in range(world)]
        recv, _ = _a2a([s[0] for s in sends], [s[1] for s in sends], 32)
        if not pipelined:
            return [x.clone() for x in recv]    # the send buffers are freed by count_records
        out = []
        for r in range(world):                  # receive into the library's own buffers, as the real driver does
            buf = ranks[r].recv_buffer(recv[r].numel() // 32)
            buf.copy_(recv[r])
            out.append(buf)
        return out

    nxt = shuffle(0) if pipelined else None
    for p in range(1 << log2_passes):
        if pipelined:
            recv = nxt
            nxt = shuffle(p + 1) if p + 1 < (1 << log2_passes) else None
        else:
            recv = shuffle(p)
        for r in range(world):
            ranks[r].count_records(recv[r], p)
            ranks[r]._n_inst_global = n_global
    return _inprocess_adjacency(ranks, world, n_global)


def _inprocess_adjacency(ranks, world, n_global):
    q = [ranks[r].adj_queries(world) for r in range(world)]
    rkeys, rcounts = _a2a([x[0] for x in q], [x[1] for x in q], 16)
    answers = [ranks[r].adj_answer(rkeys[r].contiguous()) for r in range(world)]
    back, _ = _a2a(answers, rcounts, 1)
    for r in range(world):
        ranks[r].adj_apply(back[r].contiguous())
    return n_global
