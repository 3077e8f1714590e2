// superplus_amd/csrc/df_shard.h -- the sharded createDict of `DF NUM_GPUS=N`, host side in C++: one rank per GPU drives
// libdfk's dfk_shard_* entry points and moves the records itself, over RCCL (rccl.h) directly.
//
//   reference: MapReduceEngine::Client (MapReduceEngine.h:345-388): map -> thread all-to-all -> sort/reduce, per pass
//   here:      dfk_shard_partition -> all_to_all_v of 32-byte records by owner rank (dfk_exchange.h) -> dfk_shard_count,
//              per bucket-range pass, the exchange of pass p+1 in flight under the count of pass p; then the
//              neighbour-query round trip of recomputeAdjacencies (two small all-to-alls).
// superplus_amd/dist.py is the same driver over torch.distributed and stays the test harness of the library's shard
// API; this file is what a C++ host links.
#pragma once
#include "../../include/dfk.h"
#include "dfk_exchange.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <string>
#include <thread>
#include <unistd.h>

namespace dfkx {

#define DFKX_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define DFKX_NCCL(expr) do { ncclResult_t e_ = (expr); if (e_ != ncclSuccess) throw std::runtime_error(std::string(#expr) + ": " + ncclGetErrorString(e_)); } while (0)

// RCCL over xGMI: point-to-point pieces in groups, on a stream of its own (the library counts on its own streams
// meanwhile).  The communicator's id travels through a file both sides can see (rank 0 writes it, the others wait).
struct RcclTransport : Transport {
    ncclComm_t comm = nullptr; hipStream_t st = nullptr; uint64_t* d_small = nullptr; size_t small_cap = 0;
    RcclTransport(int r, int w, int device, const std::string& id_file)
    {
        rank = r; world = w;
        DFKX_HIP(hipSetDevice(device));
        ncclUniqueId id;
        const std::string tmp = id_file + ".tmp";
        if (r == 0) {
            DFKX_NCCL(ncclGetUniqueId(&id));
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(&id, sizeof id, 1, f) != 1) throw std::runtime_error("cannot write " + tmp);
            fclose(f);
            if (rename(tmp.c_str(), id_file.c_str()) != 0) throw std::runtime_error("cannot publish " + id_file);
        } else {
            FILE* f = nullptr;
            for (int tries = 0; tries < 6000 && !(f = fopen(id_file.c_str(), "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            if (!f || fread(&id, sizeof id, 1, f) != 1) throw std::runtime_error("rank 0 never published " + id_file);
            fclose(f);
        }
        DFKX_NCCL(ncclCommInitRank(&comm, w, id, r));
        DFKX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    }
    ~RcclTransport() override { if (d_small) (void)hipFree(d_small); if (comm) (void)ncclCommDestroy(comm); if (st) (void)hipStreamDestroy(st); }
    void group_begin() override { DFKX_NCCL(ncclGroupStart()); }
    void send(const void* p, uint64_t bytes, int peer) override { DFKX_NCCL(ncclSend(p, bytes, ncclUint8, peer, comm, st)); }
    void recv(void* p, uint64_t bytes, int peer) override { DFKX_NCCL(ncclRecv(p, bytes, ncclUint8, peer, comm, st)); }
    void group_end() override { DFKX_NCCL(ncclGroupEnd()); }
    void wait() override { DFKX_HIP(hipStreamSynchronize(st)); }
    void copy_local(void* dst, const void* src, uint64_t bytes) override { DFKX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st)); }
    uint64_t* small(size_t n) { if (n > small_cap) { if (d_small) (void)hipFree(d_small); DFKX_HIP(hipMalloc(&d_small, 8 * n)); small_cap = n; } return d_small; }
    void all_reduce(uint64_t* v, int n, bool max_not_sum) override
    {
        uint64_t* d = small((size_t)n);
        DFKX_HIP(hipMemcpyAsync(d, v, 8 * (size_t)n, hipMemcpyHostToDevice, st));
        DFKX_NCCL(ncclAllReduce(d, d, (size_t)n, ncclUint64, max_not_sum ? ncclMax : ncclSum, comm, st));
        DFKX_HIP(hipMemcpyAsync(v, d, 8 * (size_t)n, hipMemcpyDeviceToHost, st));
        DFKX_HIP(hipStreamSynchronize(st));
    }
    void all_gather(const uint64_t* mine, int n, uint64_t* all) override
    {
        uint64_t* d = small((size_t)n * (world + 1));
        DFKX_HIP(hipMemcpyAsync(d, mine, 8 * (size_t)n, hipMemcpyHostToDevice, st));
        DFKX_NCCL(ncclAllGather(d, d + n, (size_t)n, ncclUint64, comm, st));
        DFKX_HIP(hipMemcpyAsync(all, d + n, 8 * (size_t)n * world, hipMemcpyDeviceToHost, st));
        DFKX_HIP(hipStreamSynchronize(st));
    }
};

struct ShardError : std::runtime_error { int code; ShardError(int c, const std::string& m) : std::runtime_error(m), code(c) {} };

struct ShardTimes { double trim = 0, plan = 0, partition = 0, exchange_wait = 0, count = 0, adjacency = 0, total = 0; uint64_t bytes_sent = 0; uint32_t n_passes = 0; };

// The sharded createDict on this rank, after dfk_shard_begin* has put the rank's reads on the device.  Afterwards the
// context holds this rank's share of the dictionary (solid sets are disjoint, spectra add).
// A library call can fail on one rank only (its own data, its own HBM budget): after every local phase the ranks
// agree on a status word, and every rank throws when any of them failed -- before the next exchange is entered.
inline void shard_create_dict(dfk_ctx* ctx, Transport& T, uint64_t n_inst_local, uint64_t piece, ShardTimes* times)
{
    using clock = std::chrono::steady_clock;
    auto secs = [](clock::time_point a) { return std::chrono::duration<double>(clock::now() - a).count(); };
    const auto t_all = clock::now();
    const int w = T.world, r = T.rank;
    int pending = 0; std::string pending_msg;
    auto note = [&](int rc) { if (rc && !pending) { pending = rc; pending_msg = dfk_last_error(); } };
    auto agree = [&](const char* what) {
        uint64_t worst = pending ? (uint64_t)(-pending) : 0;
        T.all_reduce(&worst, 1, true);
        if (pending) throw ShardError(pending, pending_msg);
        if (worst) throw ShardError(-(int)worst, std::string("another rank failed in ") + what + "; this rank stops with it");
    };
    uint64_t n_global = n_inst_local;
    T.all_reduce(&n_global, 1, false);
    if (n_global == 0) throw ShardError(DFK_E_NOGOOD, "Looks like your input data have almost no good bases.");
    auto t0 = clock::now();
    uint32_t lp = 0;
    note(dfk_shard_plan(ctx, (uint32_t)w, n_global, &lp));
    times->plan += secs(t0);
    agree("dfk_shard_plan");
    { uint64_t v = lp; T.all_reduce(&v, 1, true); lp = (uint32_t)v; }                 // every rank runs the same passes
    const uint32_t n_pass = 1u << lp;
    times->n_passes = n_pass;
    std::vector<uint64_t> rc(w), all((size_t)w * w);
    // Two passes ahead: while pass p is counted, the records of pass p+1 travel (a second receive buffer in the
    // library) and those of pass p+2 are cut by a sweep the library starts behind the count's k_count, on its second
    // stream (a second send buffer).
    struct Cut { const void* send = nullptr; std::vector<uint64_t> sc; void* recv = nullptr; uint64_t n_recv = 0; };
    auto part = [&](uint32_t p, bool defer) -> Cut {                                  // room and send counts of pass p; its kernels now or behind the next k_count
        Cut c; c.sc.assign(w, 0);
        auto t1 = clock::now();
        if (!pending) note(dfk_shard_partition_begin(ctx, (uint32_t)w, n_global, lp, p, defer ? 1 : 0, &c.send, c.sc.data()));
        times->partition += secs(t1);
        return c;
    };
    auto part_end = [&](uint32_t p) {
        auto t1 = clock::now();
        if (!pending) note(dfk_shard_partition_end(ctx, p));
        times->partition += secs(t1);
    };
    auto send_off = [&](Cut& c) {                                                     // size and start the exchange of a partitioned pass
        auto t1 = clock::now();
        T.all_gather(c.sc.data(), w, all.data());
        c.n_recv = 0;
        for (int s = 0; s < w; ++s) { rc[s] = all[(size_t)s * w + r]; c.n_recv += rc[s]; if (s != r) times->bytes_sent += 32 * c.sc[s]; }
        note(dfk_shard_recv_buffer(ctx, c.n_recv, &c.recv));
        agree("dfk_shard_recv_buffer");
        all_to_all_v(T, c.send, c.sc.data(), c.recv, rc.data(), 32, piece, /*wait=*/false);
        times->exchange_wait += secs(t1);
    };
    auto wait = [&] { auto t1 = clock::now(); T.wait(); times->exchange_wait += secs(t1); };
    Cut cur = part(0, false); part_end(0);
    agree("dfk_shard_partition");
    send_off(cur);
    wait();
    Cut ahead; bool have_ahead = false;
    if (n_pass > 1) { ahead = part(1, false); part_end(1); agree("dfk_shard_partition"); have_ahead = true; }   // (nothing to hide under yet)
    for (uint32_t p = 0; p < n_pass; ++p) {
        Cut nxt; bool sent = false;
        if (have_ahead) { nxt = ahead; send_off(nxt); sent = true; }                  // pass p+1 travels while p is counted
        have_ahead = false;
        if (p + 2 < n_pass) { ahead = part(p + 2, true); have_ahead = true; }
        auto t1 = clock::now();
        if (!pending) note(dfk_shard_count(ctx, cur.recv, cur.n_recv, p));            // a failure is agreed on below / after the loop
        times->count += secs(t1);
        if (p + 2 < n_pass) part_end(p + 2);
        if (sent) { wait(); cur = nxt; }
        if (p + 2 < n_pass) agree("dfk_shard_count / dfk_shard_partition");
    }
    std::vector<uint64_t> sc(w);
    // recomputeAdjacencies across ranks: queries to the owners, answers back in query order
    t0 = clock::now();
    const void* keys = nullptr;
    if (!pending) note(dfk_shard_adj_queries(ctx, &keys, sc.data()));
    agree("dfk_shard_count / dfk_shard_adj_queries");
    T.all_gather(sc.data(), w, all.data());
    uint64_t n_q = 0, n_in = 0;
    for (int s = 0; s < w; ++s) { rc[s] = all[(size_t)s * w + r]; n_in += rc[s]; n_q += sc[s]; if (s != r) times->bytes_sent += 17 * sc[s]; }
    // the round trip's buffers: 17 bytes per incoming query and 1 per outgoing one, beside a context that may hold most
    // of the HBM -- a rank that cannot get them says so BEFORE anybody enters the exchange (the peers would wait in it)
    void *d_in = nullptr, *d_ans = nullptr, *d_back = nullptr;
    auto room = [&](void** p, uint64_t bytes) {
        if (pending) return;
        if (hipMalloc(p, bytes) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; pending = DFK_E_NOMEM; pending_msg = "no room on the device for " + std::to_string(bytes) + " bytes of adjacency queries"; }
    };
    room(&d_in, 16 * n_in + 16); room(&d_ans, n_in + 16); room(&d_back, n_q + 16);
    try {
        agree("the adjacency buffers");
        all_to_all_v(T, keys, sc.data(), d_in, rc.data(), 16, piece);
        note(dfk_shard_adj_answer(ctx, d_in, n_in, d_ans));
        agree("dfk_shard_adj_answer");
        all_to_all_v(T, d_ans, rc.data(), d_back, sc.data(), 1, piece);
        note(dfk_shard_adj_apply(ctx, d_back, n_q));
        agree("dfk_shard_adj_apply");
    } catch (...) { (void)hipFree(d_in); (void)hipFree(d_ans); (void)hipFree(d_back); throw; }
    (void)hipFree(d_in); (void)hipFree(d_ans); (void)hipFree(d_back);
    times->adjacency += secs(t0);
    times->total = secs(t_all);
}


// After shard_create_dict: every rank's share of the dictionary to rank `root` (one all-to-all whose only non-empty
// slices point at the root), which then holds the whole dictionary and answers as after a single-GPU count
// (dfk_graph_build, dfk_paths_build).  Returns the number of solid k-mers of the whole run.
inline uint64_t shard_gather_dict(dfk_ctx* ctx, Transport& T, int root, uint64_t piece)
{
    const int w = T.world, r = T.rank;
    const void* mine = nullptr; uint64_t n_mine = 0;
    int pending = dfk_shard_dict_share(ctx, &mine, &n_mine);
    std::string pending_msg = pending ? dfk_last_error() : "";
    std::vector<uint64_t> all(w);
    T.all_gather(&n_mine, 1, all.data());
    std::vector<uint64_t> sc(w, 0), rc(w, 0);
    uint64_t total = 0, incoming = 0;
    for (int s = 0; s < w; ++s) { total += all[s]; if (r == root && s != root) { rc[s] = all[s]; incoming += all[s]; } }
    if (r != root) sc[root] = n_mine;
    void* roomp = nullptr;
    if (r == root && !pending) { pending = dfk_shard_dict_adopt(ctx, incoming, &roomp); if (pending) pending_msg = dfk_last_error(); }
    uint64_t worst = pending ? (uint64_t)(-pending) : 0;
    T.all_reduce(&worst, 1, true);
    if (pending) throw ShardError(pending, pending_msg);
    if (worst) throw ShardError(-(int)worst, "another rank failed gathering the dictionary; this rank stops with it");
    all_to_all_v(T, mine, sc.data(), roomp, rc.data(), 32, piece);
    if (r == root && dfk_shard_dict_whole(ctx)) throw ShardError(DFK_E_STATE, dfk_last_error());
    return total;
}

// After shard_create_dict: every rank's share of the dictionary to EVERY rank (world - 1 point-to-point transfers per rank, the
// same pieces as shard_gather_dict sends to one): each rank then holds the whole dictionary and answers as after a single-GPU
// count.  What the reference does with the dictionary next -- buildEdges, buildHBVFromEdges -- is 1.4 s of device and host
// work at configs[1] and deterministic (the HyperBasevector's numbering is canonical, HBVFromEdges.cc:106-111,170-238), so every
// rank builds the same graph and then paths ITS reads: nothing of the stage is left to one GPU.  99 GB at configs[2] beside a
// rank's 11 GB of reads.  Returns the number of solid k-mers of the whole run.
inline uint64_t shard_allgather_dict(dfk_ctx* ctx, Transport& T, uint64_t piece)
{
    const int w = T.world, r = T.rank;
    const void* mine = nullptr; uint64_t n_mine = 0;
    int pending = dfk_shard_dict_share(ctx, &mine, &n_mine);
    std::string pending_msg = pending ? dfk_last_error() : "";
    std::vector<uint64_t> all(w);
    T.all_gather(&n_mine, 1, all.data());
    uint64_t total = 0, incoming = 0;
    for (int s = 0; s < w; ++s) { total += all[s]; if (s != r) incoming += all[s]; }
    void* roomp = nullptr;
    if (!pending) {
        pending = dfk_shard_dict_adopt(ctx, incoming, &roomp);
        if (pending) {
            pending_msg = dfk_last_error();
            // (graph, paths, index and duplicate marks need the whole dictionary on every rank: say what the way out is)
            if (pending == DFK_E_NOMEM) {
                char more[256];
                snprintf(more, sizeof more, " -- the whole dictionary (%.1f GB) does not fit beside this rank's reads: GRAPH=False KVEC=True leaves the sharded kmers.kvec instead", 32e-9 * (double)total);
                pending_msg += more;
            }
        }
    }
    uint64_t worst = pending ? (uint64_t)(-pending) : 0;
    T.all_reduce(&worst, 1, true);
    if (pending) throw ShardError(pending, pending_msg);
    if (worst) throw ShardError(-(int)worst, "another rank failed gathering the dictionary; this rank stops with it");
    all_gather_v(T, mine, all.data(), roomp, 32, piece);
    if (dfk_shard_dict_whole(ctx)) throw ShardError(DFK_E_STATE, dfk_last_error());
    return total;
}

struct ShardPathTimes { double paths = 0, paths_write = 0, index = 0, dups = 0; uint64_t placed = 0, path_edges = 0, dup_pairs = 0; };

// Rows f-2 and f-4 on every rank (dfk.h, "rows f-2 / f-4 of a multi-GPU run"): the rank's pair range pathed and written into its
// place in a.paths; the paths index by one all-to-all of (edge, read) pairs to the owners of the edge ranges; the duplicate marks
// by one all-to-all of keys to the owners of their hash and one of answers back.  first_read / total_reads: this rank's pair
// range in the whole set.  digest[DFK_CHECK_WORDS]: the whole run's words (sums added, xors xored over the ranks).
inline void shard_paths_index_dups(dfk_ctx* ctx, Transport& T, const std::string& dir, uint64_t first_read, uint64_t total_reads, uint64_t piece,
                                   ShardPathTimes* tm, uint64_t* digest)
{
    using clock = std::chrono::steady_clock;
    auto secs = [](clock::time_point a) { return std::chrono::duration<double>(clock::now() - a).count(); };
    const int w = T.world, r = T.rank;
    int pending = 0; std::string pending_msg;
    auto note = [&](int rc) { if (rc && !pending) { pending = rc; pending_msg = dfk_last_error(); } };
    auto agree = [&](const char* what) {
        uint64_t worst = pending ? (uint64_t)(-pending) : 0;
        T.all_reduce(&worst, 1, true);
        if (pending) throw ShardError(pending, pending_msg);
        if (worst) throw ShardError(-(int)worst, std::string("another rank failed in ") + what + "; this rank stops with it");
    };
    // ---- pathReads on this rank's reads, a.paths in parts
    auto t0 = clock::now();
    note(dfk_paths_build(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, 0));
    agree("dfk_paths_build");
    tm->paths = secs(t0);
    t0 = clock::now();
    uint64_t mine[3] = {0, 0, 0};
    note(dfk_paths_var_bytes(ctx, &mine[0]));
    if (!pending) note(dfk_paths_stats(ctx, nullptr, &mine[1], &mine[2]));
    std::vector<uint64_t> all(3 * (size_t)w);
    T.all_gather(mine, 3, all.data());
    uint64_t var_before = 0, var_total = 0;
    for (int s = 0; s < w; ++s) { if (s < r) var_before += all[3 * s]; var_total += all[3 * s]; tm->placed += all[3 * s + 1]; tm->path_edges += all[3 * s + 2]; }
    if (!pending) note(dfk_paths_write_part(ctx, (dir + "/a.paths").c_str(), first_read, total_reads, var_before, var_total));
    agree("dfk_paths_write_part");
    tm->paths_write = secs(t0);
    // ---- writePathsIndex
    t0 = clock::now();
    uint64_t n_c = 0, n_v = 0, n_he = 0;
    note(dfk_graph_stats(ctx, &n_c, &n_v, &n_he));
    agree("dfk_graph_stats");
    std::vector<uint64_t> sc(w, 0), rc(w, 0), cnt(n_he, 0), sq((size_t)w * w);
    const void* pairs = nullptr;
    note(dfk_shard_pidx_pairs(ctx, (uint32_t)w, &pairs, sc.data(), cnt.data()));
    agree("dfk_shard_pidx_pairs");
    T.all_gather(sc.data(), w, sq.data());
    uint64_t n_in = 0;
    for (int s = 0; s < w; ++s) { rc[s] = sq[(size_t)s * w + r]; n_in += rc[s]; }
    if (n_he) T.all_reduce(cnt.data(), (int)n_he, false);                 // reads per edge over all ranks
    void* d_in = nullptr;
    if (hipMalloc(&d_in, 8 * n_in + 16) != hipSuccess) { (void)hipGetLastError(); d_in = nullptr; pending = DFK_E_NOMEM; pending_msg = "no room on the device for " + std::to_string(8 * n_in) + " bytes of index pairs"; }
    try {
        agree("the paths index's receive buffer");
        all_to_all_v(T, pairs, sc.data(), d_in, rc.data(), 8, piece);
        note(dfk_shard_pidx_write(ctx, (uint32_t)w, (uint32_t)r, d_in, n_in, cnt.data(), dir.c_str()));
        agree("dfk_shard_pidx_write");
    } catch (...) { (void)hipFree(d_in); throw; }
    (void)hipFree(d_in);
    tm->index = secs(t0);
    // ---- MarkDups
    t0 = clock::now();
    const void* items = nullptr;
    note(dfk_shard_dup_keys(ctx, (uint32_t)w, &items, sc.data()));
    agree("dfk_shard_dup_keys");
    T.all_gather(sc.data(), w, sq.data());
    uint64_t n_q = 0; n_in = 0;
    for (int s = 0; s < w; ++s) { rc[s] = sq[(size_t)s * w + r]; n_in += rc[s]; n_q += sc[s]; }
    void *d_items = nullptr, *d_ans = nullptr, *d_back = nullptr;
    auto room = [&](void** p, uint64_t bytes) {
        if (pending) return;
        if (hipMalloc(p, bytes + 16) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; pending = DFK_E_NOMEM; pending_msg = "no room on the device for " + std::to_string(bytes) + " bytes of duplicate keys"; }
    };
    room(&d_items, 16 * n_in); room(&d_ans, n_in); room(&d_back, n_q);
    try {
        agree("the duplicate marks' buffers");
        all_to_all_v(T, items, sc.data(), d_items, rc.data(), 16, piece);
        note(dfk_shard_dup_answer(ctx, d_items, n_in, d_ans));
        agree("dfk_shard_dup_answer");
        all_to_all_v(T, d_ans, rc.data(), d_back, sc.data(), 1, piece);
        uint64_t marked = 0;
        note(dfk_shard_dup_write(ctx, d_back, n_q, (dir + "/a.dup").c_str(), first_read / 2, total_reads / 2, &marked));
        agree("dfk_shard_dup_write");
        T.all_reduce(&marked, 1, false);
        tm->dup_pairs = marked;
    } catch (...) { (void)hipFree(d_items); (void)hipFree(d_ans); (void)hipFree(d_back); throw; }
    (void)hipFree(d_items); (void)hipFree(d_ans); (void)hipFree(d_back);
    tm->dups = secs(t0);
    // ---- the run's digests: every rank's share, sums added and xors xored
    uint64_t wds[DFK_CHECK_WORDS] = {};
    note(dfk_paths_digest(ctx, wds));
    agree("dfk_paths_digest");
    std::vector<uint64_t> every((size_t)DFK_CHECK_WORDS * w);
    T.all_gather(wds, DFK_CHECK_WORDS, every.data());
    for (int i = 0; i < DFK_CHECK_WORDS; ++i) digest[i] = 0;
    for (int s = 0; s < w; ++s) {
        const uint64_t* x = &every[(size_t)DFK_CHECK_WORDS * s];
        for (int i : {DFK_CK_PATHS_SUM, DFK_CK_N_READS, DFK_CK_N_PLACED, DFK_CK_N_PATH_EDGES, DFK_CK_INV_SUM, DFK_CK_INV_STARTS, DFK_CK_INV_ENTRIES, DFK_CK_COUNTSB_DIGEST,
                      DFK_CK_COUNTSB_SUM, DFK_CK_SELF_INVERSE, DFK_CK_DUP_DIGEST, DFK_CK_DUP_MARKED}) digest[i] += x[i];
        for (int i : {DFK_CK_PATHS_XOR, DFK_CK_INV_XOR}) digest[i] ^= x[i];
        for (int i : {DFK_CK_EDGE_KMERS, DFK_CK_N_SOLID, DFK_CK_INV_VIOLATIONS, DFK_CK_N_EDGES}) { if (s && digest[i] != x[i]) throw ShardError(DFK_E_STATE, "the ranks built different graphs from one dictionary"); digest[i] = x[i]; }
        digest[DFK_CK_VALID] = s ? (digest[DFK_CK_VALID] & x[DFK_CK_VALID]) : x[DFK_CK_VALID];
    }
}

} // namespace dfkx
