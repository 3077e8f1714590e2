// superplus_amd/csrc/dfk_exchange.h -- the k-mer shuffle of a multi-GPU run, host side, C++: one process per GPU
// calling RCCL directly (rccl.h), no Python in the path.
//
// Replaces the reference's only exchange, MapReduceEngine's thread all-to-all ("swizzle", MapReduceEngine.h:345-388):
// there thread t swaps bin (t+s+1)%T with its peer in barrier-separated rounds until every key sits with the thread
// owning hash % T; here rank r owns the minimizer buckets with (bucket & (world-1)) == r and 32-byte super-k-mer
// records travel over xGMI.
//
// The payload is NOT one ncclAllToAllv: RCCL 2.26 (ROCm 7.x) delivers only the first half of a send/recv of 2.5 GB or
// more and reports success (tools/check_rccl_large.py), and a pass of a human-scale set moves 3-7 GB between each
// pair of ranks at 2 and 4 ranks.  So: the slice a rank keeps is a device copy; every other slice travels as
// point-to-point messages of at most `piece` bytes; round j = one RCCL group holding, for every distance d = 1..world-1,
// the j-th piece to rank r+d and the j-th piece from rank r-d (both sides post in the same order; ranks need not agree
// on the number of rounds, each PAIR does).  xGMI is point-to-point, so the pieces of one round use different links.
//
// The schedule is written against a small transport interface so that it can be unit-tested without GPUs
// (tests/cpp/test_exchange.cc drives it over host memory with every rank a thread) and so that the same C++ driver runs
// with all ranks as threads of ONE process sharing one GPU (DF NUM_GPUS=n DF_TRANSPORT=loopback: the check available on
// a one-GPU box; RCCL itself refuses two ranks on one device).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <vector>

namespace dfkx {

struct Transport {
    int rank = 0, world = 1;
    virtual ~Transport() {}
    virtual void group_begin() = 0;
    virtual void send(const void* p, uint64_t bytes, int peer) = 0;          // posted; complete after group_end() + wait()
    virtual void recv(void* p, uint64_t bytes, int peer) = 0;
    virtual void group_end() = 0;
    virtual void wait() = 0;                                                  // everything posted so far has completed
    virtual void copy_local(void* dst, const void* src, uint64_t bytes) = 0; // the slice a rank keeps
    virtual void all_reduce(uint64_t* v, int n, bool max_not_sum) = 0;        // host values, every rank gets the result
    virtual void all_gather(const uint64_t* mine, int n, uint64_t* all) = 0;  // all[r*n + i] = rank r's mine[i]
};

// counts[d] units of `unit` bytes for every destination d, laid out rank after rank in `send`; what arrives is laid out
// by source rank in `recv` (recv_counts must be what the peers send: exchange them with all_gather first).
// wait == false: return once everything is posted (the caller calls T.wait() later, e.g. after counting the pass before).
inline void all_to_all_v(Transport& T, const void* send, const uint64_t* send_counts, void* recv, const uint64_t* recv_counts,
                         uint64_t unit, uint64_t piece, bool wait = true)
{
    const int w = T.world, r = T.rank;
    std::vector<uint64_t> so(w + 1, 0), ro(w + 1, 0);
    for (int i = 0; i < w; ++i) { so[i + 1] = so[i] + send_counts[i] * unit; ro[i + 1] = ro[i] + recv_counts[i] * unit; }
    if (send_counts[r] != recv_counts[r]) throw std::runtime_error("all_to_all: a rank sends itself a different amount than it expects");
    if (send_counts[r]) T.copy_local((char*)recv + ro[r], (const char*)send + so[r], send_counts[r] * unit);
    piece = std::max<uint64_t>(unit, piece / unit * unit);
    uint64_t largest = 0;
    for (int i = 0; i < w; ++i) if (i != r) largest = std::max(largest, std::max(send_counts[i], recv_counts[i]) * unit);
    for (uint64_t lo = 0; lo < largest; lo += piece) {
        T.group_begin();
        for (int d = 1; d < w; ++d) {
            const int to = (r + d) % w, from = (r - d + w) % w;
            const uint64_t sb = send_counts[to] * unit, rb = recv_counts[from] * unit;
            if (lo < sb) T.send((const char*)send + so[to] + lo, std::min(piece, sb - lo), to);
            if (lo < rb) T.recv((char*)recv + ro[from] + lo, std::min(piece, rb - lo), from);
        }
        T.group_end();
    }
    if (wait) T.wait();
}

// Every rank's `mine` (counts[rank] units) to every other rank: what arrives is laid out by source rank, THIS rank's own share
// left out (the caller already holds it), in `recv` -- the dictionary of a sharded run going to every rank (df_shard.h).  The same
// rounds of point-to-point pieces as all_to_all_v; counts[] is what every rank holds (exchange it with all_gather first).
inline void all_gather_v(Transport& T, const void* mine, const uint64_t* counts, void* recv, uint64_t unit, uint64_t piece, bool wait = true)
{
    const int w = T.world, r = T.rank;
    std::vector<uint64_t> at(w + 1, 0);
    for (int s = 0; s < w; ++s) at[s + 1] = at[s] + (s == r ? 0 : counts[s] * unit);
    piece = std::max<uint64_t>(unit, piece / unit * unit);
    uint64_t largest = 0;
    for (int s = 0; s < w; ++s) largest = std::max(largest, counts[s] * unit);
    for (uint64_t lo = 0; lo < largest; lo += piece) {
        T.group_begin();
        for (int d = 1; d < w; ++d) {
            const int to = (r + d) % w, from = (r - d + w) % w;
            const uint64_t sb = counts[r] * unit, rb = counts[from] * unit;
            if (lo < sb) T.send((const char*)mine + lo, std::min(piece, sb - lo), to);
            if (lo < rb) T.recv((char*)recv + at[from] + lo, std::min(piece, rb - lo), from);
        }
        T.group_end();
    }
    if (wait) T.wait();
}

// ---- all ranks as threads of one process: the messages are memory copies made by the receiver once both sides have
// posted.  `copy` moves the bytes (memcpy for host memory; a device-to-device copy when the ranks share a GPU).
struct LoopbackHub {
    using Copy = std::function<void(void*, const void*, uint64_t)>;
    explicit LoopbackHub(int world, Copy c = [](void* d, const void* s, uint64_t n) { memcpy(d, s, n); }) : world(world), copy(std::move(c)), box((size_t)world * world), red(world), gat(world) {}
    int world; Copy copy;
    struct Msg { const void* p; uint64_t bytes; };
    std::mutex mu; std::condition_variable cv;
    std::vector<std::vector<Msg>> box;                 // box[from * world + to]: posted sends, in order, consumed by `to`
    std::vector<size_t> taken = std::vector<size_t>((size_t)world * world, 0);
    std::vector<size_t> acked = std::vector<size_t>((size_t)world * world, 0);
    // collectives: a generation counter per kind
    std::vector<std::vector<uint64_t>> red, gat; int arrived = 0; uint64_t generation = 0; std::vector<uint64_t> result;
};

struct LoopbackTransport : Transport {
    LoopbackHub& H;
    struct Pending { bool is_send; void* p; uint64_t bytes; int peer; };
    std::vector<Pending> posted;
    std::vector<std::pair<int, size_t>> sent;          // (peer, index of my message in box[rank -> peer]) still unacknowledged
    LoopbackTransport(LoopbackHub& h, int r) : H(h) { rank = r; world = h.world; }
    void group_begin() override { posted.clear(); }
    void send(const void* p, uint64_t bytes, int peer) override { posted.push_back(Pending{true, const_cast<void*>(p), bytes, peer}); }
    void recv(void* p, uint64_t bytes, int peer) override { posted.push_back(Pending{false, p, bytes, peer}); }
    void group_end() override
    {
        {   // publish the sends
            std::lock_guard<std::mutex> g(H.mu);
            for (const Pending& q : posted) if (q.is_send) { auto& b = H.box[(size_t)rank * world + q.peer]; b.push_back(LoopbackHub::Msg{q.p, q.bytes}); sent.emplace_back(q.peer, b.size() - 1); }
        }
        H.cv.notify_all();
        for (const Pending& q : posted) {
            if (q.is_send) continue;
            const size_t slot = (size_t)q.peer * world + rank;
            LoopbackHub::Msg m;
            {
                std::unique_lock<std::mutex> g(H.mu);
                H.cv.wait(g, [&] { return H.box[slot].size() > H.taken[slot]; });
                m = H.box[slot][H.taken[slot]++];
            }
            if (m.bytes != q.bytes) throw std::runtime_error("loopback: a receive of " + std::to_string(q.bytes) + " bytes met a send of " + std::to_string(m.bytes));
            H.copy(q.p, m.p, m.bytes);
            { std::lock_guard<std::mutex> g(H.mu); ++H.acked[slot]; }
            H.cv.notify_all();
        }
        posted.clear();
    }
    void wait() override
    {   // a send is complete when its receiver has copied it (the buffer may be reused afterwards)
        std::unique_lock<std::mutex> g(H.mu);
        for (const auto& s : sent) H.cv.wait(g, [&] { return H.acked[(size_t)rank * world + s.first] > s.second; });
        sent.clear();
    }
    void copy_local(void* dst, const void* src, uint64_t bytes) override { H.copy(dst, src, bytes); }
    void rendezvous(const std::function<void()>& deposit, const std::function<void()>& combine, const std::function<void()>& collect)
    {
        std::unique_lock<std::mutex> g(H.mu);
        const uint64_t gen = H.generation;
        deposit();
        if (++H.arrived == world) { combine(); H.arrived = 0; ++H.generation; H.cv.notify_all(); }
        else H.cv.wait(g, [&] { return H.generation != gen; });
        collect();
    }
    void all_reduce(uint64_t* v, int n, bool max_not_sum) override
    {
        std::vector<uint64_t> out(n);
        rendezvous([&] { H.red[rank].assign(v, v + n); },
                   [&] { H.result.assign(n, 0); for (int i = 0; i < n; ++i) for (int r = 0; r < world; ++r) H.result[i] = max_not_sum ? std::max(H.result[i], H.red[r][i]) : H.result[i] + H.red[r][i]; },
                   [&] { out = H.result; });
        // (the result vector is rewritten by the next collective only after every rank has arrived there, i.e. after
        // every rank has left this one)
        std::copy(out.begin(), out.end(), v);
    }
    void all_gather(const uint64_t* mine, int n, uint64_t* all) override
    {
        std::vector<uint64_t> out;
        rendezvous([&] { H.gat[rank].assign(mine, mine + n); },
                   [&] { H.result.clear(); for (int r = 0; r < world; ++r) H.result.insert(H.result.end(), H.gat[r].begin(), H.gat[r].end()); },
                   [&] { out = H.result; });
        std::copy(out.begin(), out.end(), all);
    }
};

} // namespace dfkx
