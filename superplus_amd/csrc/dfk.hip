// superplus_amd/csrc/dfk.hip -- host side of libdfk.so: the C ABI of include/dfk.h over the
// HIP kernels in dfk_kernels.h.  gfx950 only; there is no CPU path in this library.
#include "../../include/dfk.h"
#include "dfk_kernels.h"

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <fcntl.h>
#include <functional>
#include <new>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

using namespace dfk;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(DFK_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// No exception crosses the C ABI: host containers can throw (a 99 GB sorted fetch asks for 200 GB of host memory).
template <class F> int guarded(F&& f)
{
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(DFK_E_NOMEM, "out of host memory"); }
    catch (const std::exception& e) { return fail(DFK_E_HIP, "internal error: %s", e.what()); }
    catch (...) { return fail(DFK_E_HIP, "internal error"); }
}

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    bool sub = false;            // carved from a pass block (dfk_ctx::PassBlock): given back with the block, not on its own
};

bool g_trace = getenv("DFK_TRACE") != nullptr;
#define TRACE(...) do { if (g_trace) { fprintf(stderr, "[dfk] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
inline double wall_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// LDS table geometry per K (DESIGN.md "count kernel"): slots and waves per workgroup
template <int K> struct CountCfg;
#ifndef DFK_LOG2S
#define DFK_LOG2S 11
#endif
#ifndef DFK_NWAVES
#define DFK_NWAVES 8
#endif
template <> struct CountCfg<40> { static constexpr int LOG2S = DFK_LOG2S, NWAVES = DFK_NWAVES; };
template <> struct CountCfg<48> { static constexpr int LOG2S = DFK_LOG2S, NWAVES = DFK_NWAVES; };
template <> struct CountCfg<60> { static constexpr int LOG2S = DFK_LOG2S, NWAVES = DFK_NWAVES; };

struct Inputs {           // device pointers
    const uint8_t* packed; uint64_t packed_bytes;
    const uint64_t* base_off; const uint32_t* read_len;
    const uint8_t* pq; uint64_t pq_bytes; const uint64_t* pq_off;
    const int32_t* bc; uint64_t n_reads;
};

} // namespace

struct dfk_ctx {
    dfk_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;             // everything except ...
    hipStream_t stream2 = nullptr;            // ... the scatter of the next pass, which runs (low priority) under the count of the current one
    hipDeviceProp_t prop{};
    uint64_t budget = 0, held = 0, peak = 0;
    struct Owned { void* p; uint64_t bytes, seq; };
    std::vector<Owned> owned;                 // live blocks of the arena, in allocation order
    uint64_t alloc_seq = 0;
    // results of the last run
    bool have = false;
    uint64_t n_reads = 0;
    struct FileRange { const char* base; uint64_t bytes; int fd; uint64_t off; };
    std::vector<FileRange> file_ranges;       // dfk_hint_file_range
    DevBuf good_len;                          // u32[n_reads]
    // dfk_entry32[n], one part per pass.  blist != 0: behind the entries, in the same buffer, 16 bytes of counters
    // and the indices of the n_blist entries that still have unresolved context bits (k_boundary_list)
    struct Part { DevBuf buf, pre; uint64_t n = 0; uint64_t blist = 0, n_blist = 0; bool listed = false; };   // listed: n_blist is all of them
    bool want_blist = false;                  // single-GPU runs: list each part's boundary entries while the next pass is counted
    int pending_blist = -1;                   // part whose list is still to be launched (see launch_boundary_list)
    std::function<int()> after_count_launch;  // the next range's sweep, when it is to start behind k_count rather than ahead of it (run_typed)
    std::vector<Part> parts;
    uint64_t n_solid = 0, n_boundary = 0;
    unsigned seg_attempt = 0;                 // the room for a pass's solid k-mers is (estimate << seg_attempt)
    double distinct_per_inst = 0.0;           // observed on the passes of the current run (sizes the work items)
    double plan_derate = 0.95;                // share of the free HBM a pass is planned into; lowered when a pass ran out (kept across runs)
    std::vector<int64_t> hist;
    std::vector<dfk_entry32> sorted, sorted_pre;
    bool sorted_ok = false, sorted_pre_ok = false;
    dfk_stats st{};
    // shard state (multi-GPU)
    bool shard_open = false, shard_use_bc = false;
    const void* sh_in[7] = {};                // caller's device arrays, valid until dfk_shard_partition returns
    uint64_t sh_packed_bytes = 0, sh_pq_bytes = 0, sh_n_reads = 0, sh_n_inst_local = 0;
    int64_t sh_read_id0 = 0;
    DevBuf shard_send[2];                     // send buffers (records grouped by destination rank) of the two passes partitioned
    uint32_t shard_send_pass[2] = {~0u, ~0u}; //   last, by pass parity: one travels while the next is being written
    DevBuf shard_recv[2];                     // receive buffers handed to the caller by dfk_shard_recv_buffer: the pass being
    unsigned shard_recv_seq = 0;              //   counted and the one being received under it alternate between the two
    uint64_t budget_taken = 0;                //   ... and the part of the budget they stand for
    void* sh_staged[6] = {};                  // dfk_shard_begin_host: this rank's inputs on the device (hipMalloc, outside the arena)
    uint64_t sh_b0 = 0, sh_b1 = 0, sh_q0 = 0, sh_q1 = 0, sh_staged_reads = 0;   //   the byte ranges of the whole set's arrays they hold, and how many reads
    DevBuf adj_keys, adj_src; uint64_t adj_n = 0;
    DevBuf set; uint64_t set_mask = 0;
    double t_upload_done = 0;                 // wall_now() when the last piece of the bases had arrived (run_under_upload)
    uint32_t shard_world = 1, shard_log2_nb = 0;
    void* shard_state = nullptr;              // bucket table + count state kept between the passes of a sharded run
    void (*shard_state_free)(void*) = nullptr;
    // transfer lanes (pinned double buffers + a stream each) are kept between transfers: a streamed a.paths is 184 batches, each a
    // transfer of its own, and pinning 8 x 4 MiB and making streams for every one of them cost more than moving the bytes
    void* lane_pool = nullptr; void (*lane_pool_free)(void*) = nullptr;
    std::mutex lane_mu;
    void* graph_state = nullptr;              // the graph built from the last count (dfk_graph.inc)
    void (*graph_state_free)(void*) = nullptr;
    // DFK_F_KEEP_INPUTS: the device copies dfk_count made of the caller's reads (hipMalloc, outside the arena), kept for
    // dfk_paths_build until the next count: packed, base_off, read_len, pq, pq_off, bc
    unsigned int* d_resident = nullptr;       // [0] workgroups of k_count that have started, ever (k_gate); [1] gates that timed out
    unsigned int resident_target = 0;         // its value once every workgroup of the launches so far has started
    unsigned int last_wg_idle = 0;            // workgroups of the last k_count launch that never counted an item (trace)
    void* kept[6] = {};
    uint64_t kept_packed_bytes = 0, kept_pq_bytes = 0, kept_n_reads = 0, kept_budget = 0;

    // Device memory comes from a few large chunks that are kept for the life of the context and managed
    // by first-fit free lists with coalescing.  hipMalloc/hipFree of multi-GB blocks cost milliseconds to
    // seconds each (and hipFree synchronises the device); a 30x human run moves hundreds of GB per pass.
    struct Free { uint64_t off, bytes; };
    struct Chunk { char* p; uint64_t bytes; std::vector<Free> free_list; bool adopted = false; };     // free_list sorted by offset; adopted: see adopt_kept
    std::vector<Chunk> chunks;
    uint64_t reserved = 0;                           // sum of chunk sizes

    uint64_t first_chunk_hint = 0;                   // set from the input size before a run: one big chunk, no growth
    // long-lived blocks (dictionary parts, goodLens, bucket counters, summaries) grow from the bottom of the chunk,
    // per-pass blocks and temporaries come from the top, so that the two kinds do not fragment each other
    bool carve(Chunk& k, size_t bytes, DevBuf& b, bool long_lived)
    {
        if (!long_lived) {                              // temporaries: the highest free block that fits, its upper end
            for (size_t i = k.free_list.size(); i-- > 0;)
                if (k.free_list[i].bytes >= bytes) {
                    k.free_list[i].bytes -= bytes;
                    b.p = k.p + k.free_list[i].off + k.free_list[i].bytes; b.bytes = bytes;
                    if (!k.free_list[i].bytes) k.free_list.erase(k.free_list.begin() + i);
                    return true;
                }
            return false;
        }
        for (size_t i = 0; i < k.free_list.size(); ++i)  // long-lived: the lowest free block that fits, its lower end
            if (k.free_list[i].bytes >= bytes) {
                b.p = k.p + k.free_list[i].off; b.bytes = bytes;
                k.free_list[i].off += bytes; k.free_list[i].bytes -= bytes;
                if (!k.free_list[i].bytes) k.free_list.erase(k.free_list.begin() + i);
                return true;
            }
        return false;
    }
    void drop_empty_chunks()
    {
        for (size_t i = 0; i < chunks.size();)
            if (chunks[i].free_list.size() == 1 && chunks[i].free_list[0].bytes == chunks[i].bytes)
            { (void)hipFree(chunks[i].p); reserved -= chunks[i].bytes; chunks.erase(chunks.begin() + i); }
            else ++i;
    }
    // Everything one pass holds while it is in flight (bucket tables, records) comes out of ONE
    // arena block, so that two passes in flight plus the growing dictionary never interleave: the blocks of
    // successive passes do not grow, so each fits the hole left by the pass before the running one.
    struct PassBlock { DevBuf block; size_t used = 0; };
    PassBlock* sub = nullptr;                        // while set, bottom allocations are bumped out of it when they fit
    int alloc(DevBuf& b, size_t bytes, const char* what, bool top = false)
    {
        bytes = bytes ? (bytes + 255) & ~(size_t)255 : 256;
        if (sub && !top && sub->used + bytes <= sub->block.bytes) {
            b.p = (char*)sub->block.p + sub->used; b.bytes = bytes; b.sub = true;
            sub->used += bytes;
            return 0;
        }
        if (held + bytes > budget)
            return fail(DFK_E_NOMEM, "HBM budget exceeded allocating %zu bytes for %s (held %llu, budget %llu)",
                        bytes, what, (unsigned long long)held, (unsigned long long)budget);
        bool ok = false;
        for (Chunk& k : chunks) if (carve(k, bytes, b, top)) { ok = true; break; }
        if (!ok) {
            // grow: the first chunk is sized from the input (a run needs a few times its input), later ones
            // twice the request (later requests reuse the slack); never past the budget
            uint64_t want = std::max<uint64_t>(2 * (uint64_t)bytes, 64ull << 20);
            if (chunks.empty()) want = std::max<uint64_t>(want, std::min<uint64_t>(first_chunk_hint, budget));
            if (reserved + want > budget) { drop_empty_chunks(); want = std::min<uint64_t>(want, budget > reserved ? budget - reserved : 0); }
            want &= ~(uint64_t)0xFFF;                  // blocks carved from the top of a chunk must stay aligned
            if (want < bytes) {
                if (g_trace) {
                    for (const Chunk& k : chunks) for (const Free& f : k.free_list) fprintf(stderr, "[dfk]   free %.2f GB at %.2f GB\n", f.bytes / 1e9, f.off / 1e9);
                    for (const Owned& o : owned) if (o.bytes >= (1ull << 28)) fprintf(stderr, "[dfk]   held %.2f GB at %.2f GB (#%llu)\n", o.bytes / 1e9, ((char*)o.p - chunks[0].p) / 1e9, (unsigned long long)o.seq);
                }
                return fail(DFK_E_NOMEM, "HBM budget exhausted by fragmentation allocating %zu bytes for %s", bytes, what);
            }
            void* p = nullptr;
            hipError_t e = hipMalloc(&p, want);
            if (e != hipSuccess && want > bytes) { (void)hipGetLastError(); want = bytes; e = hipMalloc(&p, want); }
            if (e != hipSuccess) return fail(DFK_E_NOMEM, "hipMalloc(%llu) for %s: %s", (unsigned long long)want, what, hipGetErrorString(e));
            chunks.push_back(Chunk{(char*)p, want, {Free{0, want}}});
            reserved += want;
            TRACE("new device chunk %p, %.2f GB (reserved %.2f of %.2f GB)", p, want / 1e9, (reserved) / 1e9, budget / 1e9);
            carve(chunks.back(), bytes, b, top);
        }
        held += bytes; peak = std::max(peak, held);
        owned.push_back(Owned{b.p, (uint64_t)bytes, ++alloc_seq});
        if (bytes >= (1ull << 30)) TRACE("alloc %-28s %8.2f GB at %p (%s), held %.2f GB", what, bytes / 1e9, b.p, top ? "top" : "bottom", held / 1e9);
        return 0;
    }
    // the largest block alloc() could hand out now: a free block of a chunk, or a new chunk within the budget
    uint64_t largest_allocatable() const
    {
        uint64_t best = budget > reserved ? (budget - reserved) & ~(uint64_t)0xFFF : 0;
        for (const Chunk& k : chunks) for (const Free& f : k.free_list) best = std::max<uint64_t>(best, f.bytes);
        return std::min<uint64_t>(best, budget > held ? budget - held : 0);
    }
    void release(DevBuf& b)
    {
        if (!b.p) return;
        if (b.sub) { b = DevBuf{}; return; }
        auto it = std::find_if(owned.begin(), owned.end(), [&](const Owned& o) { return o.p == b.p; });
        if (it != owned.end()) owned.erase(it);
        for (Chunk& k : chunks)
            if ((char*)b.p >= k.p && (char*)b.p < k.p + k.bytes) {
                std::vector<Free>& fl = k.free_list;
                const uint64_t off = (uint64_t)((char*)b.p - k.p);
                size_t i = 0;
                while (i < fl.size() && fl[i].off < off) ++i;
                fl.insert(fl.begin() + i, Free{off, b.bytes});
                if (i + 1 < fl.size() && fl[i].off + fl[i].bytes == fl[i + 1].off) { fl[i].bytes += fl[i + 1].bytes; fl.erase(fl.begin() + i + 1); }
                if (i > 0 && fl[i - 1].off + fl[i - 1].bytes == fl[i].off) { fl[i - 1].bytes += fl[i].bytes; fl.erase(fl.begin() + i); }
                break;
            }
        held -= b.bytes; b.p = nullptr; b.bytes = 0;
    }
    // keep only the first `keep` bytes of a block: a reservation made for an upper bound is cut to what was
    // needed; the tail goes back to the free room above it (long-lived blocks grow upwards)
    void shrink(DevBuf& b, size_t keep)
    {
        keep = keep ? (keep + 255) & ~(size_t)255 : 256;
        if (!b.p || keep >= b.bytes) return;
        auto it = std::find_if(owned.begin(), owned.end(), [&](const Owned& o) { return o.p == b.p; });
        DevBuf tail; tail.p = (char*)b.p + keep; tail.bytes = b.bytes - keep;
        b.bytes = keep;
        if (it != owned.end()) { it->bytes = keep; owned.push_back(Owned{tail.p, (uint64_t)tail.bytes, it->seq}); }
        else owned.push_back(Owned{tail.p, (uint64_t)tail.bytes, alloc_seq});
        release(tail);
    }
    // give back everything allocated after `mark` (= alloc_seq at some earlier moment): what an abandoned
    // pass left behind.  The DevBufs that pointed at those blocks are dead; the caller resets them.
    void release_since(uint64_t mark)
    {
        for (size_t i = owned.size(); i-- > 0;)
            if (owned[i].seq > mark) { DevBuf b; b.p = owned[i].p; b.bytes = owned[i].bytes; release(b); }
    }
    void drop_pool() { for (Chunk& k : chunks) (void)hipFree(k.p); chunks.clear(); reserved = 0; }
    void release_all()
    {
        // (an aborted sharded run may have left a partition registered for launch, or running on the second stream)
        after_count_launch = nullptr;
        if (stream2) (void)hipStreamSynchronize(stream2);
        // results of the previous run go back to the pool (sizes come from the DevBufs that own them)
        DevBuf* live[] = {&good_len, &shard_send[0], &shard_send[1], &shard_recv[0], &shard_recv[1], &adj_keys, &adj_src, &set};
        for (DevBuf* d : live) release(*d);
        for (Part& pt : parts) { release(pt.buf); release(pt.pre); }
        parts.clear();
        // anything an aborted run left behind: the arena is simply declared empty again
        owned.clear(); held = 0;
        for (size_t i = 0; i < chunks.size();)                                // (what adopt_kept brought in goes back to the driver: a run plans with ONE large chunk)
            if (chunks[i].adopted) { (void)hipFree(chunks[i].p); reserved -= chunks[i].bytes; chunks.erase(chunks.begin() + i); } else ++i;
        for (Chunk& k : chunks) k.free_list.assign(1, Free{0, k.bytes});
        good_len = shard_send[0] = shard_send[1] = shard_recv[0] = shard_recv[1] = adj_keys = adj_src = set = DevBuf{};
        shard_send_pass[0] = shard_send_pass[1] = ~0u;
        have = false; sorted_ok = sorted_pre_ok = false; sorted.clear(); sorted_pre.clear(); hist.clear();
        n_solid = 0; adj_n = 0; shard_open = false;
        if (shard_state) { shard_state_free(shard_state); shard_state = nullptr; }
        for (void*& p : sh_staged) if (p) { (void)hipFree(p); p = nullptr; }
        budget += budget_taken; budget_taken = 0;
        if (graph_state) { graph_state_free(graph_state); graph_state = nullptr; }
    }
    void drop_kept()
    {
        for (void*& p : kept) if (p) { (void)hipFree(p); p = nullptr; }
        budget += kept_budget; kept_budget = 0; kept_n_reads = 0;
    }
    // The kept reads' buffers become chunks of the arena instead of going back to the driver: memory that is freed is wiped
    // before anybody gets it again, at ~33 GB/s (tools/vram_alloc_cost.hip), and the step that follows the pathing (MarkDups'
    // table, 34 GB at configs[1]) asked for a new chunk right behind the release of these 91 GB -- and waited 1.5 s for it.
    void adopt_kept()
    {
        const uint64_t bytes[6] = {kept_packed_bytes + 64, (kept_n_reads + 1) * 8 + 64, kept_n_reads * 4 + 64, kept_pq_bytes + 64, (kept_n_reads + 1) * 8 + 64, 0};
        for (int i = 0; i < 6; ++i) {
            if (!kept[i]) continue;
            const uint64_t b = bytes[i] & ~(uint64_t)0xFFF;
            if (i < 5 && b >= (64ull << 20) && ((uintptr_t)kept[i] & 0xFFF) == 0) { chunks.push_back(Chunk{(char*)kept[i], b, {Free{0, b}}, true}); reserved += b; }
            else (void)hipFree(kept[i]);
            kept[i] = nullptr;
        }
        budget += kept_budget; kept_budget = 0; kept_n_reads = 0;
    }
};

namespace {

struct Timer {
    hipEvent_t a{}, b{}; hipStream_t s;
    explicit Timer(hipStream_t st) : s(st) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
    ~Timer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    void start() { (void)hipEventRecord(a, s); }
    float stop() { (void)hipEventRecord(b, s); (void)hipEventSynchronize(b); float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }
};

uint32_t ceil_log2(uint64_t v) { uint32_t b = 0; while ((1ull << b) < v) ++b; return b; }

// ------------------------------------------------------------------ stage: trim (a1)
template <int K>
int stage_trim(dfk_ctx* c, const Inputs& in, uint64_t* n_inst)
{
    DevBuf ctr; int rc = c->alloc(ctr, 32, "trim counters"); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(ctr.p, 0, 32, c->stream));
    rc = c->alloc(c->good_len, sizeof(uint32_t) * in.n_reads, "goodLens", true); if (rc) return rc;
    if (in.n_reads) {
        unsigned grid = (unsigned)std::min<uint64_t>((in.n_reads + 255) / 256, 8192);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trim<K>), dim3(grid), dim3(256), 0, c->stream,
                           in.pq, in.pq_off, in.pq_bytes, in.base_off, in.packed_bytes, in.read_len, in.n_reads, c->cfg.min_qual, (uint32_t*)c->good_len.p,
                           (unsigned long long*)ctr.p, (unsigned int*)((char*)ctr.p + 8),
                           (unsigned long long*)((char*)ctr.p + 16));
        HIP_TRY(hipGetLastError());
    }
    uint64_t h[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(h, ctr.p, 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->release(ctr);
    if ((uint32_t)h[1]) return fail(DFK_E_INPUT, "malformed input: an offset table is not monotone or runs past its array, a PQVec stream is broken, or its length differs from read_len");
    // createDict: nKmers = sum of goodLens == 0 -> "almost no good bases", Scram(1) (BuildReadQGraph48.cc:225-230)
    if (h[2] == 0) return fail(DFK_E_NOGOOD, "Looks like your input data have almost no good bases.");
    *n_inst = h[0];
    return 0;
}

// ------------------------------------------------------------------ device-side scans
// exclusive scan of n u64 on the stream (three levels cover 2^33 elements)
int device_scan(dfk_ctx* c, const uint64_t* in, uint64_t* out, uint64_t n)
{
    constexpr uint64_t PER = (uint64_t)SCAN_THREADS * SCAN_ITEMS;
    if (n == 0) return 0;
    const uint64_t nblk = (n + PER - 1) / PER;
    if (nblk == 1) {
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(SCAN_THREADS), 0, c->stream, in, out, n, (uint64_t*)nullptr);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    DevBuf sums; int rc = c->alloc(sums, nblk * 8, "scan block sums"); if (rc) return rc;
    hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)nblk), dim3(SCAN_THREADS), 0, c->stream, in, out, n, (uint64_t*)sums.p);
    HIP_TRY(hipGetLastError());
    rc = device_scan(c, (const uint64_t*)sums.p, (uint64_t*)sums.p, nblk); if (rc) return rc;
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nblk), dim3(SCAN_THREADS), 0, c->stream, out, n, (const uint64_t*)sums.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));          // sums is about to return to the pool
    c->release(sums);
    return 0;
}

// ------------------------------------------------------------------ stage: partition (a2, first half)
// Fine buckets are numbered in one global space of 2^log2_nb (owner-rank-major when sharded).  The count
// launch sees all of them once; a scatter launch materialises the records of one hash slice ("pass"):
// buckets with (id & (P-1)) == pass, renumbered id >> log2(P).  This is the reference's nPasses idea
// (MapReduceEngine.h:454) applied to HBM capacity.  Bucket tables never leave the GPU.
struct BucketTable {
    uint32_t log2_nb = 0;                 // global
    DevBuf acc;                           // u64[nb] records<<32 | instances, global numbering
    DevBuf summ;                          // uint4[n_reads]: run summaries (k_partition<K,false>), what the scatter passes work from
    DevBuf classes;                       // u32[n_reads]: coarse classes of each read's buckets (the sweeps' prefilter)
    DevBuf ovf_list; uint64_t n_ovf = 0;  // reads with too many runs for a summary: scattered by scanning
    DevBuf class_hist;                    // sharded runs: u64[2][world * PART_CLASSES] records | instances per owner and class (no per-bucket counters)
    std::vector<uint64_t> h_class;        //   ... and its host copy
    uint64_t n_records = 0, n_inst = 0;
};
struct Partition {                        // one pass
    DevBuf records, base, ipre, items;    // records; u64 base[nb+1] (first record of each bucket); u64 ipre[nb+1]
    uint64_t n_records = 0, n_inst = 0, n_items = 0;   //   (instance prefix); ItemRange items[n_items]
    uint64_t nb = 0;                      // fine buckets of this pass (world * sub_n)
};

template <int K>
PartParams part_params(const dfk_ctx* c, uint32_t log2_nb, uint32_t log2_world, int64_t read_id0, uint32_t sub_lo, uint32_t sub_n)
{
    const uint32_t M = c->cfg.minimizer_len;
    return PartParams{M, (uint32_t)K - M + 1, log2_nb, log2_world, read_id0, sub_lo, sub_n,
                      sweep_class_mask(sub_lo, sub_n, log2_nb - log2_world)};
}

// totals of a counter table (records, instances)
int table_totals(dfk_ctx* c, const DevBuf& acc, uint64_t nb, uint64_t* n_records, uint64_t* n_inst)
{
    DevBuf tot; int rc = c->alloc(tot, 16, "bucket totals"); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(tot.p, 0, 16, c->stream));
    hipLaunchKernelGGL(k_sum_acc, dim3(1024), dim3(256), 0, c->stream, (const unsigned long long*)acc.p, nb, (unsigned long long*)tot.p);
    HIP_TRY(hipGetLastError());
    uint64_t h[2];
    HIP_TRY(hipMemcpyAsync(h, tot.p, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->release(tot);
    *n_records = h[0]; *n_inst = h[1];
    return 0;
}

// The counting scan in three steps, so that a caller whose reads are still arriving (count_host: the scan runs under the upload
// of the bases) can run it a range of reads at a time: begin (tables, zeroed), range (a launch over reads [r0, r1)), end (the
// overflow list, the totals, the check against the trim's instance count).  partition_count is all three over every read.
struct ScanJob {
    PartParams pp; uint64_t nb = 0, n_bins = 0, ovf_cap = 0; bool by_class = false, ranged = false;
    DevBuf ovf_tmp, d_n;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;                  // a pair per launch: ms_part_count is their sum
    ~ScanJob() { for (auto& e : ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); } }
};

// (the register scan takes ranges; the general scan -- other minimizer lengths, DFK_OLD_SCAN -- only everything at once)
template <int K> bool scan_takes_ranges(const dfk_ctx* c)
{
    static const bool old_scan = getenv("DFK_OLD_SCAN") != nullptr;
    return c->cfg.minimizer_len == 16 && !old_scan;
}

template <int K>
int scan_begin(dfk_ctx* c, const Inputs& in, uint32_t log2_world, int64_t read_id0, BucketTable* T, bool by_class, ScanJob* J)
{
    J->pp = part_params<K>(c, T->log2_nb, log2_world, read_id0, 0, 1u << (T->log2_nb - log2_world));
    J->nb = 1ull << T->log2_nb; J->by_class = by_class;
    int rc = 0;
    J->n_bins = 2ull * (PART_CLASSES << log2_world);
    if (by_class) { rc = c->alloc(T->class_hist, J->n_bins * 8, "class counters", true); if (rc) return rc; HIP_TRY(hipMemsetAsync(T->class_hist.p, 0, J->n_bins * 8, c->stream)); }
    else { rc = c->alloc(T->acc, J->nb * 8, "bucket counters", true); if (rc) return rc; }
    rc = c->alloc(T->summ, std::max<uint64_t>(1, in.n_reads) * 16, "run summaries", true); if (rc) return rc;
    rc = c->alloc(T->classes, read_classes_bytes(std::max<uint64_t>(1, in.n_reads)), "read bucket classes", true); if (rc) return rc;
    if (!by_class) HIP_TRY(hipMemsetAsync(T->acc.p, 0, J->nb * 8, c->stream));
    // reads whose runs do not fit a summary are listed by the scan itself (two in 10^5 at 2x100 bp); if the list
    // outgrows the room set aside for it the summaries are searched instead
    J->ovf_cap = in.n_reads / 16 + 1024;
    rc = c->alloc(J->ovf_tmp, J->ovf_cap * 4, "overflow read list (scratch)"); if (rc) return rc;
    rc = c->alloc(J->d_n, 16, "overflow read count"); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(J->d_n.p, 0, 16, c->stream));
    return 0;
}

template <int K>
int scan_range(dfk_ctx* c, const Inputs& in, BucketTable* T, ScanJob* J, uint64_t r0, uint64_t r1)
{
    if (r1 <= r0) return 0;
    const PartParams& pp = J->pp;
    const bool by_class = J->by_class;
    unsigned grid = (unsigned)((r1 - r0 + PART_THREADS - 1) / PART_THREADS);
    static const unsigned scan_blocks = getenv("DFK_SCAN_BLOCKS") ? (unsigned)atoi(getenv("DFK_SCAN_BLOCKS")) : 0;
    if (by_class) grid = std::min<unsigned>(grid, (scan_blocks ? scan_blocks : 128u) * (unsigned)c->prop.multiProcessorCount);   // grid-stride: class counts are flushed once per block (12 blocks per CU: 57 ms per 225 M reads, 128: 47 ms)
    else if (scan_blocks) grid = std::min<unsigned>(grid, scan_blocks * (unsigned)c->prop.multiProcessorCount);
    const size_t lds_a = (sizeof(uint32_t) + 1) * pp.W * PART_THREADS + sizeof(uint32_t) * PART_RING * PART_THREADS + (by_class ? J->n_bins * 4 : 0);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    J->ev.emplace_back(e0, e1);
    HIP_TRY(hipEventRecord(e0, c->stream));
    // the minimizer length everybody uses gets the scan whose window lives in registers (all three K since the positions are
    // packed four to a register); other lengths the general one
    if (scan_takes_ranges<K>(c)) {
        const size_t lds_r = sizeof(uint32_t) * (PART_RING + SUMMARY_RUNS) * PART_THREADS + (by_class ? J->n_bins * 4 : 0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_count<K, 16>), dim3(grid), dim3(PART_THREADS), lds_r, c->stream,
                           in.packed, in.packed_bytes, in.base_off, (const uint32_t*)c->good_len.p, r0, r1, pp,
                           (unsigned long long*)T->acc.p, (unsigned long long*)T->class_hist.p, (unsigned long long*)J->d_n.p, J->ovf_cap,
                           (uint32_t*)J->ovf_tmp.p, (uint4*)T->summ.p, (uint32_t*)T->classes.p);
    } else {
        if (r0 != 0 || r1 != in.n_reads) return fail(DFK_E_STATE, "the general scan takes every read at once");
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_partition<K, false>), dim3(grid), dim3(PART_THREADS), lds_a, c->stream,
                           in.packed, in.packed_bytes, in.base_off, (const uint32_t*)c->good_len.p, in.bc,
                           (int64_t)c->cfg.ign_bc_below, in.n_reads, pp, (unsigned long long*)T->acc.p, (unsigned long long*)T->class_hist.p,
                           (unsigned long long*)J->d_n.p, J->ovf_cap, (uint4*)J->ovf_tmp.p, (uint4*)T->summ.p,
                           (const uint32_t*)nullptr, (uint64_t)0, (const uint64_t*)nullptr, (uint32_t*)T->classes.p);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, c->stream));
    return 0;
}

template <int K>
int scan_end(dfk_ctx* c, const Inputs& in, uint64_t n_inst, BucketTable* T, ScanJob* J)
{
    int rc = 0;
    T->n_ovf = 0;
    if (in.n_reads) {
        HIP_TRY(hipMemcpyAsync(&T->n_ovf, J->d_n.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (T->n_ovf > J->ovf_cap) {                                    // long reads: most summaries overflow
            c->release(J->ovf_tmp);
            rc = c->alloc(J->ovf_tmp, in.n_reads * 4, "overflow read list (scratch)"); if (rc) return rc;
            HIP_TRY(hipMemsetAsync(J->d_n.p, 0, 16, c->stream));
            hipLaunchKernelGGL(k_select_overflow, dim3((unsigned)std::min<uint64_t>((in.n_reads + 255) / 256, 8192)), dim3(256), 0, c->stream,
                               (const uint4*)T->summ.p, in.n_reads, (uint32_t*)J->ovf_tmp.p, (unsigned long long*)J->d_n.p);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(&T->n_ovf, J->d_n.p, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        if (T->n_ovf) {
            rc = c->alloc(T->ovf_list, T->n_ovf * 4, "overflow read list", true); if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(T->ovf_list.p, J->ovf_tmp.p, T->n_ovf * 4, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
    }
    c->release(J->ovf_tmp); c->release(J->d_n);
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0;
    for (auto& e : J->ev) { float m = 0; if (hipEventElapsedTime(&m, e.first, e.second) == hipSuccess) ms += m; }
    c->st.ms_part_count = ms; c->st.reserved[6] = J->ev.size();          // (launches of the counting scan: 1, or the pieces of run_under_upload)
    TRACE("%llu of %llu reads have more than %d runs", (unsigned long long)T->n_ovf, (unsigned long long)in.n_reads, SUMMARY_RUNS);
    if (J->by_class) {
        T->h_class.assign(J->n_bins, 0);
        HIP_TRY(hipMemcpyAsync(T->h_class.data(), T->class_hist.p, J->n_bins * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        T->n_records = T->n_inst = 0;
        for (uint64_t i = 0; i < J->n_bins / 2; ++i) { T->n_records += T->h_class[i]; T->n_inst += T->h_class[J->n_bins / 2 + i]; }
    } else { rc = table_totals(c, T->acc, J->nb, &T->n_records, &T->n_inst); if (rc) return rc; }
    TRACE("partition count pass done (%llu buckets, %llu records%s)", (unsigned long long)J->nb, (unsigned long long)T->n_records, J->ev.size() > 1 ? ", in ranges under the upload" : "");
    if (T->n_inst != n_inst)
        return fail(DFK_E_HIP, "partition count pass saw %llu instances, trim saw %llu%s",
                    (unsigned long long)T->n_inst, (unsigned long long)n_inst,
                    (n_inst - T->n_inst) % (1ull << 32) == 0 ? " (a fine bucket with 2^32 or more instances: its counter keeps 32 bits)" : "");
    c->st.n_records = T->n_records; c->st.n_buckets = J->nb;
    return 0;
}

template <int K>
int partition_count(dfk_ctx* c, const Inputs& in, uint64_t n_inst, uint32_t log2_world, int64_t read_id0, BucketTable* T, bool by_class = false)
{
    ScanJob J;
    int rc = scan_begin<K>(c, in, log2_world, read_id0, T, by_class, &J); if (rc) return rc;
    rc = scan_range<K>(c, in, T, &J, 0, in.n_reads); if (rc) return rc;
    return scan_end<K>(c, in, n_inst, T, &J);
}

// base[], ipre[] and the work items of one pass, all on the device
int pass_tables(dfk_ctx* c, const DevBuf& acc, uint32_t log2_sub, uint32_t world, uint32_t sub_lo, uint32_t sub_n, uint64_t budget, Partition* P)
{
    P->nb = (uint64_t)world * sub_n;
    const uint64_t nb = P->nb;
    DevBuf rec, inst, flags, idx;
    int rc = c->alloc(rec, (nb + 1) * 8, "bucket record counts"); if (rc) return rc;
    rc = c->alloc(inst, (nb + 1) * 8, "bucket instance counts"); if (rc) return rc;
    rc = c->alloc(P->base, (nb + 1) * 8, "bucket bases"); if (rc) return rc;
    rc = c->alloc(P->ipre, (nb + 1) * 8, "bucket instance prefix"); if (rc) return rc;
    const unsigned g1 = (unsigned)((nb + 1 + 255) / 256);
    hipLaunchKernelGGL(k_slice, dim3(g1), dim3(256), 0, c->stream, (const unsigned long long*)acc.p, log2_sub, sub_lo, sub_n, nb,
                       (uint64_t*)rec.p, (uint64_t*)inst.p);
    HIP_TRY(hipGetLastError());
    rc = device_scan(c, (const uint64_t*)rec.p, (uint64_t*)P->base.p, nb + 1); if (rc) return rc;
    rc = device_scan(c, (const uint64_t*)inst.p, (uint64_t*)P->ipre.p, nb + 1); if (rc) return rc;
    // items
    rc = c->alloc(flags, (nb + 1) * 8, "item flags"); if (rc) return rc;
    rc = c->alloc(idx, (nb + 1) * 8, "item index"); if (rc) return rc;
    hipLaunchKernelGGL(k_item_flags, dim3(g1), dim3(256), 0, c->stream, (const uint64_t*)P->ipre.p, nb, budget, (uint64_t*)flags.p);
    HIP_TRY(hipGetLastError());
    rc = device_scan(c, (const uint64_t*)flags.p, (uint64_t*)idx.p, nb + 1); if (rc) return rc;
    uint64_t tot[3];
    HIP_TRY(hipMemcpyAsync(&tot[0], (const uint64_t*)P->base.p + nb, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&tot[1], (const uint64_t*)P->ipre.p + nb, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&tot[2], (const uint64_t*)idx.p + nb, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    P->n_records = tot[0]; P->n_inst = tot[1]; P->n_items = tot[2];
    rc = c->alloc(P->items, std::max<uint64_t>(1, P->n_items) * sizeof(ItemRange), "work items"); if (rc) return rc;
    hipLaunchKernelGGL(k_item_pairs, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream, (const uint64_t*)flags.p,
                       (const uint64_t*)idx.p, nb, (ItemRange*)P->items.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->release(rec); c->release(inst); c->release(flags); c->release(idx);
    return 0;
}

void release_pass(dfk_ctx* c, Partition* P)
{ c->release(P->records); c->release(P->base); c->release(P->ipre); c->release(P->items); }

// Instances per work item: 1.5 x the table's slots to start with; once a pass has shown how many distinct
// k-mers an instance brings (0.25 at K=48 and 58x, more at lower coverage or larger K), the next passes aim at
// tables a good third full, which keeps the share of items that overflow their table near 0.3 %.
template <int K> uint64_t item_budget(const dfk_ctx* c)
{
    constexpr uint64_t S = 1ull << CountCfg<K>::LOG2S;
    if (c->cfg.inst_per_item) return c->cfg.inst_per_item;
    if (c->distinct_per_inst <= 0.0) return 3 * S / 2;
    const double b = 0.367 * (double)S / c->distinct_per_inst;
    return (uint64_t)std::min(std::max(b, 0.5 * (double)S), 3.0 * (double)S);
}

// One pass's records: begin() builds the pass's bucket tables and enqueues the scatter on c->stream without
// waiting for it; end() waits, checks that every bucket received what the counting scan saw, and frees the
// cursors.  (run_typed enqueues the next pass's scatter on a second stream before counting the current one.)
struct ScatterJob { Partition P; DevBuf cur, d_bad; hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; uint32_t lo = 0, n = 0; };

template <int K>
int scatter_launch(dfk_ctx* c, const Inputs& in, const BucketTable& T, uint32_t log2_world, int64_t read_id0, ScatterJob* J)
{
    const PartParams pp = part_params<K>(c, T.log2_nb, log2_world, read_id0, J->lo, J->n);
    HIP_TRY(hipEventRecord(J->e0, J->st));
    if (in.n_reads)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scatter_runs<K>), dim3((unsigned)((in.n_reads + 256ull * sweep_reads<K>() - 1) / (256ull * sweep_reads<K>()))),
                           dim3(256), 0, J->st,
                           in.packed, in.packed_bytes, in.base_off, (const uint32_t*)c->good_len.p, in.bc,
                           (int64_t)c->cfg.ign_bc_below, in.n_reads, pp, (const uint4*)T.summ.p, (const uint32_t*)T.classes.p,
                           (unsigned long long*)J->cur.p, J->P.n_records, (uint4*)J->P.records.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(J->e1, J->st));
    return 0;
}

template <int K>
int scatter_begin(dfk_ctx* c, const Inputs& in, const BucketTable& T, uint32_t log2_world, int64_t read_id0,
                  uint32_t sub_lo, uint32_t sub_n, ScatterJob* J, bool launch_now = true)
{
    Partition* P = &J->P;
    J->st = c->stream; J->lo = sub_lo; J->n = sub_n;
    const PartParams pp = part_params<K>(c, T.log2_nb, log2_world, read_id0, sub_lo, sub_n);
    const uint64_t budget = item_budget<K>(c);
    int rc = pass_tables(c, T.acc, T.log2_nb - log2_world, 1u << log2_world, sub_lo, sub_n, budget, P); if (rc) return rc;
    const uint64_t nb = P->nb;
    rc = c->alloc(J->cur, nb * 8, "bucket cursors"); if (rc) return rc;
    rc = c->alloc(P->records, P->n_records * 32, "super-k-mer records"); if (rc) return rc;
    rc = c->alloc(J->d_bad, 16, "scatter check"); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(J->cur.p, P->base.p, nb * 8, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(J->d_bad.p, 0, 16, c->stream));
    if (!J->e0) { HIP_TRY(hipEventCreate(&J->e0)); HIP_TRY(hipEventCreate(&J->e1)); }
    (void)pp;
    if (launch_now) return scatter_launch<K>(c, in, T, log2_world, read_id0, J);
    return 0;
}

template <int K>
int scatter_end(dfk_ctx* c, const Inputs& in, const BucketTable& T, uint32_t log2_world, int64_t read_id0, ScatterJob* J)
{
    HIP_TRY(hipEventSynchronize(J->e1));
    float ms = 0; (void)hipEventElapsedTime(&ms, J->e0, J->e1);
    c->st.ms_part_scatter += ms;
    // The few reads with more runs than a summary holds are scanned again -- here, on the main stream, between
    // two counts: a handful of blocks on the low-priority stream wait milliseconds for a wave slot while
    // k_count runs.  Then every bucket must have received exactly the records counted for it.
    {
        Partition* P = &J->P;
        const PartParams pp = part_params<K>(c, T.log2_nb, log2_world, read_id0, J->lo, J->n);
        const size_t lds_b = sizeof(uint32_t) * pp.W * PART_THREADS + sizeof(uint32_t) * 2 * PART_QCAP * PART_THREADS;
        Timer t(c->stream);
        t.start();
        if (T.n_ovf)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_partition<K, true>), dim3((unsigned)((T.n_ovf + PART_THREADS - 1) / PART_THREADS)),
                               dim3(PART_THREADS), lds_b, c->stream,
                               in.packed, in.packed_bytes, in.base_off, (const uint32_t*)c->good_len.p, in.bc,
                               (int64_t)c->cfg.ign_bc_below, in.n_reads, pp, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                               (unsigned long long*)J->cur.p, P->n_records, (uint4*)P->records.p, (uint4*)nullptr,
                               (const uint32_t*)T.ovf_list.p, T.n_ovf, (const uint64_t*)nullptr, (uint32_t*)nullptr);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_check_cursors, dim3((unsigned)std::min<uint64_t>((P->nb + 255) / 256, 4096)), dim3(256), 0, c->stream, (const unsigned long long*)J->cur.p,
                           (const uint64_t*)P->base.p, P->nb, (unsigned int*)J->d_bad.p);
        HIP_TRY(hipGetLastError());
        c->st.ms_part_scatter += t.stop();
    }
    unsigned int bad = 0;
    HIP_TRY(hipMemcpy(&bad, J->d_bad.p, 4, hipMemcpyDeviceToHost));
    c->release(J->d_bad); c->release(J->cur);
    if (bad) return fail(DFK_E_HIP, "scatter: %u buckets did not receive the records counted for them (run summaries and bucket counters disagree)", bad);
    TRACE("scatter of buckets [%u, %u) done (%llu records, %llu items)", J->lo, J->lo + J->n,
          (unsigned long long)J->P.n_records, (unsigned long long)J->P.n_items);
    return 0;
}

// ------------------------------------------------------------------ stage: count (a2 second half, a3, a4, a5)
constexpr int E_SPLIT_NO_ROOM = -101;  // internal: count_split could not even start (no room for the expanded records): sub-passes instead
constexpr int E_SEGMENT_FULL = -100;   // internal: the room reserved for a pass's solid k-mers was too small; redo the pass with more

struct CountRun {                     // device state shared by the count launches of one run
    CountGlobals* g = nullptr; uint4* seg = nullptr; unsigned long long* hist = nullptr;
    CountParams cp{}; unsigned grid = 0;
    DevBuf big; uint64_t big_cap = 0;  // output of the HBM-table fallback (its own buffer)
    DevBuf d_hist, d_g;                // spectrum bins and counters: live across the passes of one run
    DevBuf d_snap;                     // their state before the current pass (a pass that runs out of room is undone and redone)
    DevBuf d_part;                     // room reserved for the dense part of the pass being counted (count_prepare .. count_run)
    DevBuf d_wg;                       // WgOut of every persistent workgroup
    uint64_t solid_seen = 0, inst_seen = 0;   // totals of the passes done so far (sizes the next pass's output)
    uint64_t boundary_seen = 0;               // entries with unresolved context bits emitted by the passes done so far
    uint64_t splits_seen = 0;                 // items cut in two inside k_count by the passes done so far
};

int count_run_begin(dfk_ctx* c, CountRun* R)
{
    int rc = c->alloc(R->d_hist, (uint64_t)HIST_GLOBAL_BINS * 8, "spectrum bins", true); if (rc) return rc;
    rc = c->alloc(R->d_g, sizeof(CountGlobals), "count globals", true); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(R->d_hist.p, 0, (uint64_t)HIST_GLOBAL_BINS * 8, c->stream));
    HIP_TRY(hipMemsetAsync(R->d_g.p, 0, sizeof(CountGlobals), c->stream));
    R->g = (CountGlobals*)R->d_g.p; R->hist = (unsigned long long*)R->d_hist.p;
    return c->alloc(R->d_snap, (uint64_t)HIST_GLOBAL_BINS * 8 + 256, "spectrum snapshot", true);
}

int count_snapshot(dfk_ctx* c, CountRun* R, bool restore)
{
    char* snap = (char*)R->d_snap.p;
    const uint64_t hb = (uint64_t)HIST_GLOBAL_BINS * 8;
    if (!restore) {
        HIP_TRY(hipMemcpyAsync(snap, R->d_hist.p, hb, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(snap + hb, R->d_g.p, sizeof(CountGlobals), hipMemcpyDeviceToDevice, c->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(R->d_hist.p, snap, hb, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(R->d_g.p, snap + hb, sizeof(CountGlobals), hipMemcpyDeviceToDevice, c->stream));
    }
    return 0;
}

// spectrum (a5): bins 0..max count, after the last pass
int count_run_end(dfk_ctx* c, CountRun* R)
{
    CountGlobals hg{};
    HIP_TRY(hipMemcpy(&hg, R->d_g.p, sizeof hg, hipMemcpyDeviceToHost));
    c->st.n_distinct = hg.n_distinct; c->n_boundary = hg.n_boundary;
#ifdef DFK_PHASE_TIMES
    {
        unsigned long long ph[16] = {};
        (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof ph);
        unsigned long long tot = 0; for (int i = 0; i < 12; ++i) tot += ph[i];
        static const char* nm[12] = {"item loop tail (thread 0 publishes)", "barrier: item start", "chunks (stage + insert)", "barrier: chunks done", "finish pass 1 (solidity)", "barrier", "finish pass 2 (adjacency look-ups)", "barrier", "finish pass 3 (emit)", "barrier", "clear + reduce", "drain"};
        fprintf(stderr, "[dfk] k_count wave cycles by phase (%.3e memtime ticks in all):\n", (double)tot);
        for (int i = 0; i < 12; ++i) fprintf(stderr, "[dfk]   %5.1f %%  %s\n", 100.0 * (double)ph[i] / (double)std::max(1ull, tot), nm[i]);
        unsigned long long z[16] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z);
    }
#endif
#ifdef DFK_PROBE_STATS
    {
        unsigned long long ps[4] = {};
        (void)hipMemcpyFromSymbol(ps, HIP_SYMBOL(g_probe_stats), sizeof ps);
        fprintf(stderr, "[dfk] probe stats: %.2f loop iterations per batch, %.3f probes and %.3f lock waits per instance (%llu batches)\n",
                (double)ps[0] / (double)std::max(1ull, ps[1]), (double)ps[2] / (64.0 * std::max(1ull, ps[1])),
                (double)ps[3] / (64.0 * std::max(1ull, ps[1])), ps[1]);
        unsigned long long z[4] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_probe_stats), z, sizeof z);
    }
#endif
    DevBuf d_max; int rc = c->alloc(d_max, 16, "max bin"); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d_max.p, 0, 16, c->stream));
    hipLaunchKernelGGL(k_hist_max, dim3(1024), dim3(256), 0, c->stream, (const unsigned long long*)R->d_hist.p, HIST_GLOBAL_BINS,
                       (unsigned int*)d_max.p);
    uint32_t nb = 0;
    HIP_TRY(hipMemcpyAsync(&nb, d_max.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->hist.assign(nb, 0);
    if (nb) HIP_TRY(hipMemcpy(c->hist.data(), R->d_hist.p, 8ull * nb, hipMemcpyDeviceToHost));
    TRACE("spectrum read back: %u bins, %llu solid", nb, (unsigned long long)c->n_solid);
    c->release(d_max); c->release(R->d_hist); c->release(R->d_g); c->release(R->d_snap);
    return 0;
}

template <int K> unsigned count_grid(const dfk_ctx* c, int nbc = 1)      // persistent workgroups of k_count: as many as fit the LDS
{
    constexpr int LOG2S = CountCfg<K>::LOG2S, NW = CountCfg<K>::NWAVES;
    const size_t lds = count_lds_bytes<K, LOG2S, NW, 1>() + (size_t)(nbc > 1 ? nbc - 1 : 0) * sizeof(uint32_t) * (1u << LOG2S);   // one more word per slot and barcode beyond the first
    const unsigned per_cu = (unsigned)std::max<size_t>(1, (size_t)(160 * 1024) / lds);
    return (unsigned)c->prop.multiProcessorCount * per_cu;
}

// The boundary list of the part finished last is launched on the second stream once the NEXT pass's k_count is
// running: launched right after its own pass it streams the part (12 GB) through the window between two counts,
// where the small kernels of the main stream (overflow reads, cursor check) then take 1-2 ms each instead of 0.2.
int launch_boundary_list(dfk_ctx* c)
{
    if (c->pending_blist < 0) return 0;
    const dfk_ctx::Part& part = c->parts[(size_t)c->pending_blist];
    c->pending_blist = -1;
    unsigned long long* ctl = (unsigned long long*)((char*)part.buf.p + part.blist);
    HIP_TRY(hipMemsetAsync(ctl, 0, 16, c->stream2));
    hipLaunchKernelGGL(k_boundary_list, dim3((unsigned)std::min<uint64_t>((part.n + 2047) / 2048, 4096)), dim3(256), 0, c->stream2,
                       (const uint4*)part.buf.p, part.n, (uint32_t*)(ctl + 2), part.n_blist, ctl);
    HIP_TRY(hipGetLastError());
    return 0;
}

// one k_count launch over `n_items` device-resident items; overflowed items come back as bucket ranges
template <int K, int NBC>
int launch_count(dfk_ctx* c, const Partition& P, const ItemRange* d_items, uint64_t n_items, const CountRun& R,
                 std::vector<ItemRange>* overflowed, float* kernel_ms, const uint32_t* d_sub = nullptr, bool single_kmer_records = false)
{
    constexpr int LOG2S = CountCfg<K>::LOG2S, NW = CountCfg<K>::NWAVES;
    if (n_items == 0) return 0;
    if (n_items >= (1ull << 32)) return fail(DFK_E_ARG, "too many work items in one pass");
    DevBuf d_ovf;
    int rc = c->alloc(d_ovf, n_items * sizeof(ItemRange), "overflow list"); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(&R.g->next_item, 0, 8, c->stream));      // next_item, n_overflow
    HIP_TRY(hipMemsetAsync(&R.g->n_wg_idle, 0, 4, c->stream));
    CountParams cp = R.cp; cp.n_items = (uint32_t)n_items; cp.single = single_kmer_records ? 1u : 0u;
    const size_t lds = count_lds_bytes<K, LOG2S, NW, NBC>();
    auto kern = d_sub ? k_count<K, LOG2S, NW, NBC, true> : k_count<K, LOG2S, NW, NBC, false>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    TRACE("k_count: %llu items, grid %u, lds %zu, seg_cap %llu", (unsigned long long)n_items, R.grid, lds, (unsigned long long)R.cp.seg_cap);
    Timer tk(c->stream);
    tk.start();
    hipLaunchKernelGGL(kern, dim3(R.grid), dim3(NW * 64), lds, c->stream,
                       (const uint4*)P.records.p, d_items, (const uint64_t*)P.base.p, cp, R.g, R.seg, (WgOut*)R.d_wg.p, R.hist,
                       (ItemRange*)d_ovf.p, d_sub, c->d_resident);
    HIP_TRY(hipGetLastError());
    c->resident_target += R.grid;
    if (c->after_count_launch || c->pending_blist >= 0) {
        // what the second stream is about to run beside this launch waits until its persistent workgroups are all in place
        // (at most 2 ms: then it goes ahead and the gate says so)
        static const bool no_gate = getenv("DFK_NO_GATE") != nullptr;
        if (!no_gate) hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, c->stream2, (const unsigned int*)c->d_resident, c->resident_target, 200000ull, c->d_resident + 1);
    }
    if (c->after_count_launch) { std::function<int()> f; f.swap(c->after_count_launch); rc = f(); if (rc) return rc; }
    rc = launch_boundary_list(c); if (rc) return rc;
    *kernel_ms += tk.stop();
    CountGlobals g{};
    HIP_TRY(hipMemcpyAsync(&g, R.g, sizeof g, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->last_wg_idle = g.n_wg_idle;
    if (g.n_overflow) {
        const size_t at = overflowed->size();
        overflowed->resize(at + g.n_overflow);
        HIP_TRY(hipMemcpy(overflowed->data() + at, d_ovf.p, sizeof(ItemRange) * g.n_overflow, hipMemcpyDeviceToHost));
    }
    c->release(d_ovf);
    return 0;
}

template <int K, int NBC>
int launch_count_big(dfk_ctx* c, const Partition& P, const std::vector<ItemRange>& singles, CountRun& R)
{
    constexpr int KW = KTraits<K>::KW, NW = 8;
    // instance counts of the single buckets (ipre[b1] - ipre[b0]) size their tables: >= 2x the instances,
    // an upper bound on the distinct k-mers
    const uint32_t n = (uint32_t)singles.size();
    std::vector<uint32_t> idx(2 * n);
    for (uint32_t i = 0; i < n; ++i) { idx[2 * i] = singles[i].b0; idx[2 * i + 1] = singles[i].b1; }
    DevBuf d_idx, d_val;
    int rc = c->alloc(d_idx, 8ull * n, "gather index"); if (rc) return rc;
    rc = c->alloc(d_val, 16ull * n, "gather values"); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_idx.p, idx.data(), 8ull * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_gather_u64, dim3((2 * n + 255) / 256), dim3(256), 0, c->stream, (const uint64_t*)P.ipre.p,
                       (const uint32_t*)d_idx.p, 2 * n, (uint64_t*)d_val.p);
    HIP_TRY(hipGetLastError());
    std::vector<uint64_t> val(2 * n);
    HIP_TRY(hipMemcpyAsync(val.data(), d_val.p, 16ull * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->release(d_idx); c->release(d_val);
    std::vector<BigItem> items; uint64_t words = 0, tot_inst = 0;
    std::vector<uint64_t> chunk_pre(n + 1, 0), slot_pre(n + 1, 0), rec(2 * n);
    // record ranges of the items (for the chunk tickets)
    {
        DevBuf d_i2, d_v2;
        rc = c->alloc(d_i2, 8ull * n, "gather index"); if (rc) return rc;
        rc = c->alloc(d_v2, 16ull * n, "gather values"); if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(d_i2.p, idx.data(), 8ull * n, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_gather_u64, dim3((2 * n + 255) / 256), dim3(256), 0, c->stream, (const uint64_t*)P.base.p,
                           (const uint32_t*)d_i2.p, 2 * n, (uint64_t*)d_v2.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(rec.data(), d_v2.p, 16ull * n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->release(d_i2); c->release(d_v2);
    }
    for (uint32_t i = 0; i < n; ++i) {
        tot_inst += val[2 * i + 1] - val[2 * i];
        chunk_pre[i + 1] = chunk_pre[i] + (rec[2 * i + 1] - rec[2 * i] + COUNT_CHUNK - 1) / COUNT_CHUNK;
    }
    // A table is sized for the distinct k-mers its bucket is expected to hold -- instances x (distinct k-mers per
    // instance seen so far, with a margin) -- at load <= 0.5; 2 x instances is the certain bound and costs 56 bytes
    // per instance (42 GB for the hot buckets of one pass of a human-scale set with a 10 % repeat family).  An insert
    // that runs out of probe steps says the guess was too low: every table is then rebuilt twice as large.
    double per_inst = c->distinct_per_inst > 0.0 ? std::min(1.0, std::max(0.05, 1.5 * c->distinct_per_inst)) : 1.0;
    DevBuf d_items, d_fail, d_pre;
    {
        // the fallback's own output buffer; a pass can come here more than once (the sub-buckets of split hot buckets,
        // then the pass's other buckets): what an earlier call emitted stays in front
        const uint64_t more = tot_inst / std::max<uint32_t>(1, c->cfg.min_freq) + 1;   // every solid k-mer has >= min_freq instances
        if (!R.big.p) { R.big_cap = more; rc = c->alloc(R.big, R.big_cap * 32, "fallback solid entries"); if (rc) return rc; }
        else {
            CountGlobals g0{};
            HIP_TRY(hipMemcpy(&g0, R.d_g.p, sizeof g0, hipMemcpyDeviceToHost));
            DevBuf grown;
            rc = c->alloc(grown, (g0.big_cursor + more) * 32, "fallback solid entries"); if (rc) return rc;
            if (g0.big_cursor) HIP_TRY(hipMemcpy(grown.p, R.big.p, g0.big_cursor * 32, hipMemcpyDeviceToDevice));
            c->release(R.big);
            R.big = grown; R.big_cap = g0.big_cursor + more;
        }
    }
    rc = c->alloc(d_items, (uint64_t)n * sizeof(BigItem), "fallback items"); if (rc) return rc;
    rc = c->alloc(d_pre, 16ull * (n + 1), "fallback prefixes"); if (rc) return rc;
    rc = c->alloc(d_fail, 16, "fallback flag"); if (rc) return rc;
    uint64_t* d_chunk_pre = (uint64_t*)d_pre.p; uint64_t* d_slot_pre = d_chunk_pre + (n + 1);
    HIP_TRY(hipMemcpyAsync(d_chunk_pre, chunk_pre.data(), 8ull * (n + 1), hipMemcpyHostToDevice, c->stream));
    CountParams cpb = R.cp; cpb.seg_cap = R.big_cap;                       // the fallback writes to its own buffer
    const unsigned cus = (unsigned)c->prop.multiProcessorCount;
    for (;;) {
        items.clear(); words = 0;
        bool certain = true;
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t inst = val[2 * i + 1] - val[2 * i];
            const uint64_t guess = std::min<uint64_t>(inst, (uint64_t)((double)inst * per_inst) + 256);
            certain = certain && guess == inst;
            const uint32_t l2 = std::max<uint32_t>(13, ceil_log2(2 * guess + 64));
            items.push_back(BigItem{singles[i].b0, singles[i].b1, words, l2, 0});
            words += (uint64_t)(KW + 4 + (NBC > 1 ? NBC - 1 : 0)) << l2;   // keys, state, contexts, counts, barcode words (BigView)
            slot_pre[i + 1] = slot_pre[i] + (1ull << l2);
        }
        DevBuf pool;
        rc = c->alloc(pool, words * 4 + 8, "HBM fallback tables"); if (rc) return rc;
        // (zeroed by our own grid-stride kernel: the pool of a pass with a 10^8-instance bucket is past 4 GiB)
        hipLaunchKernelGGL(k_fill_u64, dim3(4096), dim3(256), 0, c->stream, (uint64_t*)pool.p, words / 2 + (words & 1), 0ull);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemsetAsync(d_fail.p, 0, 16, c->stream));
        HIP_TRY(hipMemcpyAsync(d_items.p, items.data(), items.size() * sizeof(BigItem), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_slot_pre, slot_pre.data(), 8ull * (n + 1), hipMemcpyHostToDevice, c->stream));
        // the whole grid works on the fallback tables together (d_fail + 8: the chunk ticket)
        const unsigned g_ins = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((chunk_pre[n] + NW * BIG_TICKET_CHUNKS - 1) / (NW * BIG_TICKET_CHUNKS), 4ull * cus));
        const unsigned g_slot = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((slot_pre[n] + 255) / 256, 16ull * cus));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_big_insert<K, NW, NBC>), dim3(g_ins), dim3(NW * 64), 0, c->stream,
                           (const uint4*)P.records.p, (const BigItem*)d_items.p, (const uint64_t*)P.base.p, (const uint64_t*)d_chunk_pre, n,
                           (uint32_t*)pool.p, (unsigned long long*)d_fail.p + 1, (uint32_t*)d_fail.p);
        HIP_TRY(hipGetLastError());
        uint32_t failed = 0;
        HIP_TRY(hipMemcpyAsync(&failed, d_fail.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (failed) {
            unsigned int info[4] = {};
            (void)hipMemcpyFromSymbol(info, HIP_SYMBOL(g_big_fail), sizeof info);
            const unsigned int zero[4] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_big_fail), zero, sizeof zero);
            c->release(pool);
            if (certain) {
                c->release(d_items); c->release(d_fail); c->release(d_pre);
                return fail(DFK_E_HIP, "an insert into an HBM fallback table gave up after %u probe steps and %u waits on a locked slot "
                                       "(table of 2^%u slots at load <= 0.5; %u tables, %llu instances in this pass's fallback)", info[1], info[2], info[3], n,
                            (unsigned long long)tot_inst);
            }
            per_inst = std::min(1.0, 2.0 * per_inst);
            TRACE("fallback: an HBM table filled up (2^%u slots): rebuilding the tables for %.2f distinct k-mers per instance", info[3], per_inst);
            continue;
        }
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_big_flags<K, NBC>), dim3(g_slot), dim3(256), 0, c->stream, (const BigItem*)d_items.p,
                           (const uint64_t*)d_slot_pre, n, (uint32_t*)pool.p, cpb, R.g);
        if (cpb.do_adj)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_big_resolve<K>), dim3(g_slot), dim3(256), 0, c->stream, (const BigItem*)d_items.p,
                               (const uint64_t*)d_slot_pre, n, (uint32_t*)pool.p);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_big_emit<K>), dim3(g_slot), dim3(256), 0, c->stream, (const BigItem*)d_items.p,
                           (const uint64_t*)d_slot_pre, n, (uint32_t*)pool.p, cpb, R.g, (uint4*)R.big.p, R.hist);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->release(pool);
        break;
    }
    c->release(d_items); c->release(d_fail); c->release(d_pre);
    return 0;
}

// Fine buckets too rich for one LDS table, the linear way: one pass over their records writes every instance out as
// a record of one k-mer, grouped by a second hash of the canonical k-mer into sub-buckets of ~700 distinct k-mers
// (k_hot_split: count, scan, scatter); the sub-buckets are then an ordinary small pass for k_count.  What still does
// not fit (a guess too low) goes to the HBM tables.  Returns DFK_E_NOMEM untouched when the expanded records do not
// fit: the caller then falls back to sub-passes.
template <int K, int NBC>
int count_split(dfk_ctx* c, const Partition& P, const std::vector<ItemRange>& buckets, const std::vector<uint64_t>& inst,
                const std::vector<uint32_t>& log2p, CountRun& R)
{
    constexpr int NW = 8;
    const uint32_t n = (uint32_t)buckets.size();
    std::vector<HotItem> items(n);
    std::vector<uint32_t> idx(2 * n);
    uint64_t ns = 0, tot = 0;
    for (uint32_t i = 0; i < n; ++i) { items[i] = HotItem{buckets[i].b0, buckets[i].b1, (uint32_t)ns, log2p[i]}; ns += 1ull << log2p[i]; tot += inst[i]; idx[2 * i] = buckets[i].b0; idx[2 * i + 1] = buckets[i].b1; }
    if (ns >= (1ull << 24)) { fail(DFK_E_NOMEM, "more than 2^24 sub-buckets in one pass"); return E_SPLIT_NO_ROOM; }   // (the record header keeps 24 bits)
    const uint64_t mark = c->alloc_seq;
    bool started = false;             // once a sub-bucket has been counted there is no way back to sub-passes: an error undoes the pass
    auto undo = [&](int rc) {
        (void)hipStreamSynchronize(c->stream);
        if (!started) { c->release_since(mark); return rc == DFK_E_NOMEM ? E_SPLIT_NO_ROOM : rc; }
        return rc;
    };
    // record ranges of the buckets -> chunk tickets
    DevBuf d_idx, d_val, d_items, d_pre, d_tk, acc;
    int rc = c->alloc(d_idx, 8ull * n, "gather index"); if (rc) return undo(rc);
    rc = c->alloc(d_val, 16ull * n, "gather values"); if (rc) return undo(rc);
    HIP_TRY(hipMemcpyAsync(d_idx.p, idx.data(), 8ull * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_gather_u64, dim3((2 * n + 255) / 256), dim3(256), 0, c->stream, (const uint64_t*)P.base.p, (const uint32_t*)d_idx.p, 2 * n, (uint64_t*)d_val.p);
    HIP_TRY(hipGetLastError());
    std::vector<uint64_t> rec(2 * n), chunk_pre(n + 1, 0);
    HIP_TRY(hipMemcpyAsync(rec.data(), d_val.p, 16ull * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < n; ++i) chunk_pre[i + 1] = chunk_pre[i] + (rec[2 * i + 1] - rec[2 * i] + COUNT_CHUNK - 1) / COUNT_CHUNK;
    rc = c->alloc(d_items, (uint64_t)n * sizeof(HotItem), "hot buckets"); if (rc) return undo(rc);
    rc = c->alloc(d_pre, 8ull * (n + 1), "hot bucket chunks"); if (rc) return undo(rc);
    rc = c->alloc(d_tk, 16, "hot bucket ticket"); if (rc) return undo(rc);
    rc = c->alloc(acc, 8ull * ns, "sub-bucket counters"); if (rc) return undo(rc);
    HIP_TRY(hipMemcpyAsync(d_items.p, items.data(), (uint64_t)n * sizeof(HotItem), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_pre.p, chunk_pre.data(), 8ull * (n + 1), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(d_tk.p, 0, 16, c->stream));
    HIP_TRY(hipMemsetAsync(acc.p, 0, 8ull * ns, c->stream));
    const unsigned cus = (unsigned)c->prop.multiProcessorCount;
    // counting pass (sub-bucket counters in LDS, HOT_BLOCK chunks per workgroup ticket)
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((chunk_pre[n] + HOT_BLOCK - 1) / HOT_BLOCK, 2ull * cus));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_hot_pass<K, NW, false>), dim3(grid), dim3(NW * 64), 0, c->stream, (const uint4*)P.records.p, (const HotItem*)d_items.p,
                       (const uint64_t*)P.base.p, (const uint64_t*)d_pre.p, n, (unsigned long long*)d_tk.p, (unsigned long long*)acc.p,
                       (unsigned long long*)nullptr, (const uint64_t*)nullptr, (uint64_t)0, (uint4*)nullptr);
    HIP_TRY(hipGetLastError());
    Partition P2;
    rc = pass_tables(c, acc, 0, 1, 0, (uint32_t)ns, item_budget<K>(c), &P2); if (rc) return undo(rc);
    if (P2.n_inst != tot || P2.n_records != tot) return undo(fail(DFK_E_HIP, "hot buckets: %llu instances split, %llu expected", (unsigned long long)P2.n_inst, (unsigned long long)tot));
    DevBuf cur;
    rc = c->alloc(cur, 8ull * ns, "sub-bucket cursors"); if (rc) return undo(rc);
    rc = c->alloc(P2.records, 32ull * tot, "hot buckets' k-mer records"); if (rc) return undo(rc);
    HIP_TRY(hipMemcpyAsync(cur.p, P2.base.p, 8ull * ns, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(d_tk.p, 0, 16, c->stream));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_hot_pass<K, NW, true>), dim3(grid), dim3(NW * 64), 0, c->stream, (const uint4*)P.records.p, (const HotItem*)d_items.p,
                       (const uint64_t*)P.base.p, (const uint64_t*)d_pre.p, n, (unsigned long long*)d_tk.p, (unsigned long long*)nullptr,
                       (unsigned long long*)cur.p, (const uint64_t*)P2.base.p, tot, (uint4*)P2.records.p);
    HIP_TRY(hipGetLastError());
    {   // every sub-bucket must have received exactly what the counting pass saw for it
        HIP_TRY(hipMemsetAsync(d_tk.p, 0, 16, c->stream));
        hipLaunchKernelGGL(k_check_cursors, dim3((unsigned)std::min<uint64_t>((ns + 255) / 256, 4096)), dim3(256), 0, c->stream, (const unsigned long long*)cur.p,
                           (const uint64_t*)P2.base.p, ns, (unsigned int*)d_tk.p);
        uint32_t bad = 0;
        HIP_TRY(hipMemcpyAsync(&bad, d_tk.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (bad) return undo(fail(DFK_E_HIP, "hot buckets: %u of %llu sub-buckets did not receive the instances counted for them", bad, (unsigned long long)ns));
    }
    TRACE("fallback: %u hot buckets, %llu instances written out as %llu sub-buckets (%llu items)", n, (unsigned long long)tot, (unsigned long long)ns, (unsigned long long)P2.n_items);
    // the sub-buckets as a pass of their own
    started = true;
    c->st.n_items += P2.n_items;
    std::vector<ItemRange> overflowed, singles;
    float ignored = 0;
    rc = launch_count<K, NBC>(c, P2, (const ItemRange*)P2.items.p, P2.n_items, R, &overflowed, &ignored, nullptr, true); if (rc) return undo(rc);
    while (!overflowed.empty()) {
        std::vector<ItemRange> next;
        for (const ItemRange& r : overflowed) {
            if (r.b1 - r.b0 <= 1) { singles.push_back(r); continue; }
            const uint32_t mid = r.b0 + (r.b1 - r.b0) / 2;
            next.push_back({r.b0, mid}); next.push_back({mid, r.b1});
        }
        overflowed.clear();
        if (next.empty()) break;
        DevBuf d_next; rc = c->alloc(d_next, next.size() * sizeof(ItemRange), "split items"); if (rc) return undo(rc);
        HIP_TRY(hipMemcpyAsync(d_next.p, next.data(), next.size() * sizeof(ItemRange), hipMemcpyHostToDevice, c->stream));
        rc = launch_count<K, NBC>(c, P2, (const ItemRange*)d_next.p, next.size(), R, &overflowed, &ignored, nullptr, true);
        if (rc) return undo(rc);
    }
    TRACE("fallback: %u buckets (%llu instances) split into %llu sub-buckets, %llu items; %zu sub-buckets to HBM tables", n, (unsigned long long)tot,
          (unsigned long long)ns, (unsigned long long)P2.n_items, singles.size());
    if (!singles.empty()) { rc = launch_count_big<K, NBC>(c, P2, singles, R); if (rc) return undo(rc); }
    HIP_TRY(hipStreamSynchronize(c->stream));
    // (everything allocated here goes back except R.big, the HBM path's output, which count_run collects)
    const DevBuf keep_big = R.big;
    for (size_t i = c->owned.size(); i-- > 0;)
        if (c->owned[i].seq > mark && c->owned[i].p != keep_big.p) { DevBuf b; b.p = c->owned[i].p; b.bytes = c->owned[i].bytes; c->release(b); }
    return 0;
}

// Count one pass: run k_count over its items (+ split / HBM-table fallbacks), gather the pass's solid
// k-mers into a dense part.
// barcodes a table slot remembers: max(1, MIN_BC - 1); 0 without barcodes
int barcode_words(const dfk_ctx* c, bool have_bc)
{ return !have_bc ? 0 : (int)std::max<uint32_t>(1, std::min<uint32_t>(c->cfg.min_bc, DFK_MAX_MIN_BC) - (c->cfg.min_bc > 1 ? 1 : 0)); }

// solid k-mers a pass of n_inst instances is expected to emit at most (what its part's reservation is sized for)
uint64_t solid_cap(const dfk_ctx* c, const CountRun& R, uint64_t n_inst)
{
    uint64_t cap = n_inst / std::max<uint32_t>(1, c->cfg.min_freq) + 1;
    // (buckets are hash-distributed: once a pass has been counted the solid/instance ratio holds to a fraction of a percent)
    if (R.inst_seen) cap = std::min<uint64_t>(cap, (uint64_t)(1.12 * (double)R.solid_seen / (double)R.inst_seen * (double)n_inst) + 65536);
    else cap = std::min<uint64_t>(cap, n_inst / 10 + (1u << 20));   // first pass: a prior (30x data: n_inst/15, 58x: n_inst/28); too small -> redone
    return cap;
}

template <int K> constexpr uint64_t out_chunk() { return 2ull << CountCfg<K>::LOG2S; }   // = table_finish's OUT_CHUNK

template <int K>
int count_prepare(dfk_ctx* c, const Partition& P, CountRun& R, int nbc)
{
    const unsigned attempt = c->seg_attempt;
    // Output: the pass's part of the dictionary, reserved now (from the bottom of the arena's free room, where the
    // dictionary grows) so that what is placed next -- the block of the following range -- cannot fragment the
    // room it needs.  The persistent workgroups fill it chunk by chunk (WgOut); count_run cuts it to size.
    // Every solid k-mer has >= min_freq instances, which bounds the total; after the first pass the observed
    // solid/instance ratio holds to a fraction of a percent (buckets are hash-distributed).
    R.grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(P.n_items, count_grid<K>(c, nbc)));
    uint64_t res = solid_cap(c, R, P.n_inst);
    if (R.inst_seen) res = std::min<uint64_t>(res, (uint64_t)(1.10 * (double)R.solid_seen / (double)R.inst_seen * (double)P.n_inst) + 65536);
    res = (res << attempt) + (uint64_t)R.grid * out_chunk<K>();       // + the chunk ends the workgroups leave empty
    const uint64_t room = c->budget > c->held ? (c->budget - c->held) : 0;
    if (res * 32 > room) res = room / 32;
    int rc = c->alloc(R.d_wg, sizeof(WgOut) * R.grid, "workgroup output state"); if (rc) return rc;
    std::vector<WgOut> init(R.grid, WgOut{~0ull, (unsigned int)out_chunk<K>(), 0u});   // no chunk yet: the first item takes one
    HIP_TRY(hipMemcpy(R.d_wg.p, init.data(), sizeof(WgOut) * R.grid, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(&R.g->big_cursor, 0, 8, c->stream));
    HIP_TRY(hipMemsetAsync(&R.g->part_cursor, 0, 8, c->stream));
    R.cp = CountParams{c->cfg.min_freq, c->cfg.min_bc, 0, 0, res, c->cfg.min_freq > 1 ? 1u : 0u,
                       (c->cfg.flags & DFK_F_KEEP_PRE_ADJ) ? 1u : 0u};
    R.big_cap = 0;
    rc = c->alloc(R.d_part, std::max<uint64_t>(1, res) * 32, "solid k-mer entries", true); if (rc) return rc;
    R.seg = (uint4*)R.d_part.p;
    return 0;
}

template <int K, int NBC>
int count_run(dfk_ctx* c, const Partition& P, CountRun& R)
{
    int rc = 0;
    Timer t(c->stream);
    std::vector<ItemRange> overflowed;
    c->st.n_items += P.n_items;
    const float ms_before = c->st.ms_count;
    rc = launch_count<K, NBC>(c, P, (const ItemRange*)P.items.p, P.n_items, R, &overflowed, &c->st.ms_count); if (rc) return rc;
    if (g_trace) {
        unsigned int gate[2] = {0, 0};
        (void)hipMemcpy(gate, c->d_resident, 8, hipMemcpyDeviceToHost);
        TRACE("k_count: %llu instances in %.1f ms (%.1f G instances/s), reservation %.2f GB, %u of %u workgroups idle, %u gate timeouts so far",
              (unsigned long long)P.n_inst, c->st.ms_count - ms_before, 1e-6 * (double)P.n_inst / (double)(c->st.ms_count - ms_before), R.d_part.bytes / 1e9,
              c->last_wg_idle, R.grid, gate[1]);
    }
    t.start();
    // items that overflowed their LDS table are halved by bucket index and retried; a single fine bucket
    // that still overflows is counted in an HBM table
    c->st.n_overflow_items += overflowed.size();
    std::vector<ItemRange> singles;
    while (!overflowed.empty()) {
        std::vector<ItemRange> next;
        for (const ItemRange& r : overflowed) {
            if (r.b1 - r.b0 <= 1) { singles.push_back(r); continue; }
            const uint32_t mid = r.b0 + (r.b1 - r.b0) / 2;
            next.push_back({r.b0, mid}); next.push_back({mid, r.b1});
        }
        overflowed.clear();
        if (next.empty()) break;
        TRACE("retrying %zu split items", next.size());
        DevBuf d_next; rc = c->alloc(d_next, next.size() * sizeof(ItemRange), "split items"); if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(d_next.p, next.data(), next.size() * sizeof(ItemRange), hipMemcpyHostToDevice, c->stream));
        float ignored = 0;
        rc = launch_count<K, NBC>(c, P, (const ItemRange*)d_next.p, next.size(), R, &overflowed, &ignored);
        c->release(d_next);
        if (rc) return rc;
    }
    TRACE("fallback: %zu single-bucket items", singles.size());
    if (!singles.empty()) {
        // A fine bucket too rich for one LDS table is counted in 2^p sub-passes of k_count, each taking the k-mers
        // of one selector value (all instances of a k-mer share it, so solidity and counts are exact; neighbours in
        // another sub-pass are settled with the other cross-item bits).  p from the bucket's instance count, which
        // bounds its distinct k-mers: 1024 per sub-pass at most, in a 2048-slot table that gives up at 1536.
        // Buckets beyond 64 sub-passes (a minimizer owning a sizeable share of the genome) get an HBM table.
        const uint32_t n = (uint32_t)singles.size();
        std::vector<uint32_t> idx(2 * n);
        for (uint32_t i = 0; i < n; ++i) { idx[2 * i] = singles[i].b0; idx[2 * i + 1] = singles[i].b1; }
        DevBuf d_idx, d_val;
        rc = c->alloc(d_idx, 8ull * n, "gather index"); if (rc) return rc;
        rc = c->alloc(d_val, 16ull * n, "gather values"); if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(d_idx.p, idx.data(), 8ull * n, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_gather_u64, dim3((2 * n + 255) / 256), dim3(256), 0, c->stream, (const uint64_t*)P.ipre.p,
                           (const uint32_t*)d_idx.p, 2 * n, (uint64_t*)d_val.p);
        HIP_TRY(hipGetLastError());
        std::vector<uint64_t> val(2 * n);
        HIP_TRY(hipMemcpyAsync(val.data(), d_val.p, 16ull * n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->release(d_idx); c->release(d_val);
        std::vector<ItemRange> sub_items, huge; std::vector<uint32_t> sub_words;
        constexpr uint64_t PER_SUB = (1ull << CountCfg<K>::LOG2S) / 2;
        constexpr uint32_t MAX_P = 8;                                    // selector bits (wave_count_chunk)
        static const uint32_t max_lds_p = getenv("DFK_MAX_SUBPASS_LOG2") ? (uint32_t)atoi(getenv("DFK_MAX_SUBPASS_LOG2")) : 8;
        // p is first guessed from the distinct k-mers per instance seen so far (a repeat-rich bucket has far fewer
        // distinct k-mers than instances); a sub-pass that overflows anyway is cut in two by one more selector bit
        // and counted again -- its siblings are done and stay -- until p = MAX_P, where 1024 *instances* per sub-pass
        // are guaranteed.
        const double dpi = c->distinct_per_inst > 0.0 ? std::min(1.0, 2.0 * c->distinct_per_inst) : 0.5;   // (first pass: a guess)
        // Buckets that would need four sub-passes or more (each sub-pass reads and extracts ALL of the bucket's
        // instances again: work quadratic in the bucket's size, 6 s of a 9.5 s step at human scale with a 10 % repeat
        // family) are partitioned a second time instead, by k-mer hash (count_split): linear work.
        static const uint32_t split_from_p = getenv("DFK_SPLIT_FROM_LOG2") ? (uint32_t)atoi(getenv("DFK_SPLIT_FROM_LOG2")) : 2;
        std::vector<ItemRange> sp_b; std::vector<uint64_t> sp_i; std::vector<uint32_t> sp_p; std::vector<uint32_t> sp_at;
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t inst = val[2 * i + 1] - val[2 * i];
            const uint64_t guess = (uint64_t)((double)inst * dpi) + 1;
            const uint32_t p = ceil_log2((guess + 699) / 700);
            // (k_hot_pass keeps a record's place inside its hot bucket in 32 bits: a bucket of 2^32 instances or more -- whatever
            // its distinct k-mers -- takes the HBM tables)
            if (p >= split_from_p && p <= 22 && inst < (1ull << 32)) { sp_b.push_back(singles[i]); sp_i.push_back(inst); sp_p.push_back(p); sp_at.push_back(i); }
        }
        std::vector<uint8_t> taken(n, 0);
        // The expanded records of the split buckets (32 B per instance: 30 GB for a pass of a human-scale set with a 10 %
        // repeat family) must fit one free block of the arena, which the pass plan does not reserve: the buckets are
        // split in as many groups as that takes.  (All at once or not at all, three passes in twenty found no block
        // and fell back to 250 000 sub-passes and HBM tables: 0.45 s each.)
        for (size_t i0 = 0; i0 < sp_b.size();) {
            const uint64_t can = c->largest_allocatable(), keep = 512ull << 20;
            const uint64_t cap_inst = can > keep ? (uint64_t)(0.95 * (double)(can - keep)) / 34 : 0;
            size_t i1 = i0; uint64_t sum = 0;
            while (i1 < sp_b.size() && sum + sp_i[i1] <= cap_inst) sum += sp_i[i1++];
            if (i1 == i0) { TRACE("fallback: no room to split a hot bucket of %llu instances (%.2f GB free in one piece): sub-passes instead", (unsigned long long)sp_i[i0], can / 1e9); break; }
            const std::vector<ItemRange> gb(sp_b.begin() + i0, sp_b.begin() + i1);
            const std::vector<uint64_t> gi(sp_i.begin() + i0, sp_i.begin() + i1);
            const std::vector<uint32_t> gp(sp_p.begin() + i0, sp_p.begin() + i1);
            const int r2 = count_split<K, NBC>(c, P, gb, gi, gp, R);
            if (r2 == 0) { for (size_t i = i0; i < i1; ++i) taken[sp_at[i]] = 1; i0 = i1; }
            else if (r2 != E_SPLIT_NO_ROOM) return r2;
            else { TRACE("fallback: no room to split %zu hot buckets (%s): sub-passes instead", i1 - i0, g_err.c_str()); break; }
        }
        for (uint32_t i = 0; i < n; ++i) {
            if (taken[i]) continue;
            const uint64_t inst = val[2 * i + 1] - val[2 * i];
            // (beyond MAX_P selector bits x 1024 instances the refinement below could not be guaranteed to end)
            if (ceil_log2((inst + PER_SUB - 1) / PER_SUB) > MAX_P) { huge.push_back(singles[i]); continue; }
            const uint64_t guess = (uint64_t)((double)inst * dpi) + 1;
            const uint32_t p = std::max<uint32_t>(1, ceil_log2((guess + PER_SUB - 1) / PER_SUB));
            // Every sub-pass reads and extracts ALL of the bucket's instances again: 2^p-fold work, quadratic in the
            // bucket's size -- at human scale with a 10 % repeat family the sub-passes take 6 s of a 9.5 s step.  The
            // HBM tables are no way out as they stand: k_big_insert runs at 1.8 G instances/s there (four dependent
            // agent-scope atomics per instance, acquire/release fences per probe: 245 ms per pass), 8.7 s per step when
            // every bucket beyond four sub-passes goes to them (DFK_MAX_SUBPASS_LOG2=2).  What such buckets want is a
            // second-level partition by k-mer hash into buckets that fit LDS tables.
            if (p > max_lds_p) { huge.push_back(singles[i]); continue; }
            for (uint32_t k = 0; k < (1u << p); ++k) { sub_items.push_back(singles[i]); sub_words.push_back((p << 8) | k); }
        }
        TRACE("fallback: %zu sub-passes over %zu buckets in LDS tables, %zu buckets in HBM tables", sub_items.size(), singles.size() - huge.size(), huge.size());
        while (!sub_items.empty()) {
            DevBuf d_it, d_sub;
            rc = c->alloc(d_it, sub_items.size() * sizeof(ItemRange), "sub-pass items"); if (rc) return rc;
            rc = c->alloc(d_sub, sub_words.size() * 4, "sub-pass words"); if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(d_it.p, sub_items.data(), sub_items.size() * sizeof(ItemRange), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_sub.p, sub_words.data(), sub_words.size() * 4, hipMemcpyHostToDevice, c->stream));
            std::vector<ItemRange> again;
            float ignored = 0;
            rc = launch_count<K, NBC>(c, P, (const ItemRange*)d_it.p, sub_items.size(), R, &again, &ignored, (const uint32_t*)d_sub.p);
            c->release(d_it); c->release(d_sub);
            if (rc) return rc;
            sub_items.clear(); sub_words.clear();
            for (const ItemRange& r : again) {                           // {bucket, 0x80000000 | sub-pass word}
                const uint32_t w = r.b1 & 0x7FFFFFFFu, p = w >> 8, k = w & 0xFFu;
                if (!(r.b1 & 0x80000000u) || p >= MAX_P)
                    return fail(DFK_E_HIP, "a sub-pass (%u of 2^%u) of bucket %u overflowed its table (it holds at most %llu instances)", k, p, r.b0, (unsigned long long)PER_SUB);
                sub_items.push_back(ItemRange{r.b0, r.b0 + 1}); sub_words.push_back(((p + 1) << 8) | k);
                sub_items.push_back(ItemRange{r.b0, r.b0 + 1}); sub_words.push_back(((p + 1) << 8) | (k + (1u << p)));
            }
            if (!sub_items.empty()) TRACE("fallback: %zu sub-passes cut in two", again.size());
        }
        if (!huge.empty()) { rc = launch_count_big<K, NBC>(c, P, huge, R); if (rc) return rc; }
    }
    c->st.ms_fallback += t.stop();

    CountGlobals hg{};
    std::vector<WgOut> wg(R.grid);
    HIP_TRY(hipMemcpy(&hg, R.d_g.p, sizeof hg, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(wg.data(), R.d_wg.p, sizeof(WgOut) * R.grid, hipMemcpyDeviceToHost));
    c->release(R.d_wg);
    const uint64_t CH = out_chunk<K>(), claimed = hg.part_cursor;
    // the holes: the unused end of every workgroup's last chunk
    std::vector<std::pair<uint64_t, uint64_t>> holes;
    uint64_t n_holes = 0;
    for (const WgOut& w : wg) if (w.chunk != ~0ull && w.used < CH) { holes.push_back({w.chunk + w.used, w.chunk + CH}); n_holes += CH - w.used; }
    const uint64_t n_lds = claimed - n_holes;                          // solid k-mers the LDS path emitted
    if (hg.solid_overflow || claimed > R.cp.seg_cap || hg.big_cursor > R.big_cap || n_lds + hg.big_cursor > R.cp.seg_cap) {
        // the caller undoes the pass and redoes it with more room
        c->release(R.big); c->release(R.d_part);
        fail(DFK_E_NOMEM, "the room reserved for a pass's solid k-mers (%llu entries) is full", (unsigned long long)R.cp.seg_cap);
        return E_SEGMENT_FULL;
    }
    // entries beyond n_lds move into the holes below n_lds: afterwards [0, n_lds) is dense
    std::sort(holes.begin(), holes.end());
    std::vector<uint64_t> dst, src;
    for (const auto& h : holes) for (uint64_t i = h.first; i < std::min(h.second, n_lds); ++i) dst.push_back(i);
    {
        size_t hi = 0;
        for (uint64_t i = n_lds; i < claimed && src.size() < dst.size(); ++i) {
            while (hi < holes.size() && holes[hi].second <= i) ++hi;
            if (hi < holes.size() && holes[hi].first <= i) { i = holes[hi].second - 1; continue; }   // skip a hole
            src.push_back(i);
        }
    }
    if (src.size() != dst.size())
        return fail(DFK_E_HIP, "output compaction: %zu holes, %zu entries to move", dst.size(), src.size());
    dfk_ctx::Part part;
    part.n = n_lds + hg.big_cursor;
    part.buf = R.d_part; R.d_part = DevBuf{};
    if (!src.empty()) {
        DevBuf d_mv; rc = c->alloc(d_mv, 16ull * src.size(), "hole moves"); if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(d_mv.p, src.data(), 8ull * src.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync((uint64_t*)d_mv.p + src.size(), dst.data(), 8ull * src.size(), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_fill_holes, dim3((unsigned)((src.size() + 255) / 256)), dim3(256), 0, c->stream, (uint4*)part.buf.p,
                           (const uint64_t*)d_mv.p, (const uint64_t*)d_mv.p + src.size(), (uint64_t)src.size());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->release(d_mv);
    }
    if (hg.big_cursor)
        HIP_TRY(hipMemcpyAsync((char*)part.buf.p + 32 * n_lds, R.big.p, 32 * hg.big_cursor, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->release(R.big);
    // the part's boundary list goes behind its entries if the reservation has the room (it almost always has:
    // 4 bytes for one entry in eight); it is filled on the second stream while the next pass is counted
    const uint64_t nb_part = hg.n_boundary - R.boundary_seen;
    R.boundary_seen = hg.n_boundary;
    c->st.n_overflow_items += hg.n_split - R.splits_seen;             // (+ the single buckets that went to HBM tables, above)
    R.splits_seen = hg.n_split;
    uint64_t keep = part.n * 32;
    part.listed = c->want_blist && nb_part == 0;
    if (c->want_blist && nb_part && keep + 16 + 4 * nb_part <= part.buf.bytes && part.n < (1ull << 32)) {
        part.blist = keep; part.n_blist = nb_part; part.listed = true;
        keep += 16 + 4 * nb_part;
    }
    c->shrink(part.buf, keep);
    c->parts.push_back(part);
    if (part.blist) c->pending_blist = (int)c->parts.size() - 1;     // launched under the next pass's count (launch_boundary_list)
    c->n_solid += part.n; c->st.n_solid = c->n_solid;
    R.solid_seen += part.n; R.inst_seen += P.n_inst;
    if (R.inst_seen) c->distinct_per_inst = (double)hg.n_distinct / (double)R.inst_seen;
    return 0;
}


template <int K>
int stage_count(dfk_ctx* c, const Partition& P, CountRun& R, bool have_bc)
{
    int rc = count_prepare<K>(c, P, R, barcode_words(c, have_bc)); if (rc) return rc;
    switch (barcode_words(c, have_bc)) {
    case 0: return count_run<K, 0>(c, P, R);
    case 1: return count_run<K, 1>(c, P, R);
    case 2: return count_run<K, 2>(c, P, R);
    case 3: return count_run<K, 3>(c, P, R);
    case 4: return count_run<K, 4>(c, P, R);
    case 5: return count_run<K, 5>(c, P, R);
    case 6: return count_run<K, 6>(c, P, R);
    default: return count_run<K, 7>(c, P, R);
    }
}

// ------------------------------------------------------------------ stage: adjacency (a6)
// k_count has already settled every context bit whose neighbour was counted in the same table.  What is
// left (pad byte 0 of an entry) concerns neighbours in other items: those k-mers go into an HBM set.
int build_set(dfk_ctx* c)
{
    uint64_t slots = 1ull << std::max<uint32_t>(10, ceil_log2(2 * c->n_boundary + 2));
    int rc = c->alloc(c->set, slots * sizeof(SetSlot), "boundary k-mer set"); if (rc) return rc;
    c->set_mask = slots - 1;
    hipLaunchKernelGGL(k_fill_u64, dim3(2048), dim3(256), 0, c->stream, (uint64_t*)c->set.p, slots * 2, ~0ull);
    for (const dfk_ctx::Part& pt : c->parts)
        if (pt.n)
            hipLaunchKernelGGL(k_set_insert, dim3((unsigned)((pt.n + 255) / 256)), dim3(256), 0, c->stream,
                               (const uint4*)pt.buf.p, pt.n, (SetSlot*)c->set.p, c->set_mask);
    HIP_TRY(hipGetLastError());
    return 0;
}

// DFK_F_KEEP_PRE_ADJ: the kmers.kvec view (contexts before any clean-up)
int make_pre_view(dfk_ctx* c)
{
    if (!(c->cfg.flags & DFK_F_KEEP_PRE_ADJ)) return 0;
    for (dfk_ctx::Part& pt : c->parts) {
        int rc = c->alloc(pt.pre, pt.n * 32, "pre-adjacency copy"); if (rc) return rc;
        if (!pt.n) continue;
        if (c->cfg.min_freq > 1)
            hipLaunchKernelGGL(k_make_pre, dim3((unsigned)((pt.n + 255) / 256)), dim3(256), 0, c->stream,
                               (uint4*)pt.buf.p, (uint4*)pt.pre.p, pt.n);
        else
            HIP_TRY(hipMemcpyAsync(pt.pre.p, pt.buf.p, pt.n * 32, hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int K>
int stage_adjacency(dfk_ctx* c)
{
    Timer t(c->stream);
    t.start();
    int rc = make_pre_view(c); if (rc) return rc;
    if (c->cfg.min_freq > 1 && c->n_solid && c->n_boundary) {          // BuildReadQGraph48.cc:313
        // every part listed its boundary entries (k_boundary_list, second stream)?  Then the set is built and
        // queried from the lists; otherwise by streaming the parts.
        bool listed = c->want_blist;
        for (const dfk_ctx::Part& pt : c->parts) listed = listed && pt.listed;
        if (c->want_blist) {                                            // the last part's list is still to be made
            rc = launch_boundary_list(c); if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(c->stream2));
        }
        if (listed) {
            uint64_t total = 0;
            for (const dfk_ctx::Part& pt : c->parts) {
                if (!pt.blist) continue;
                unsigned long long ctl[2] = {0, 0};
                HIP_TRY(hipMemcpy(ctl, (const char*)pt.buf.p + pt.blist, 16, hipMemcpyDeviceToHost));
                if (ctl[1] || ctl[0] != pt.n_blist)
                    return fail(DFK_E_HIP, "boundary list: %llu entries listed, the count kernels reported %llu", ctl[0], (unsigned long long)pt.n_blist);
                total += pt.n_blist;
            }
            // (parts without unresolved bits have no list; the total must be the run's)
            if (total != c->n_boundary) listed = false;
        }
        if (!listed) { rc = build_set(c); if (rc) return rc; }
        else {
            uint64_t slots = 1ull << std::max<uint32_t>(10, ceil_log2(2 * c->n_boundary + 2));
            rc = c->alloc(c->set, slots * sizeof(SetSlot), "boundary k-mer set"); if (rc) return rc;
            c->set_mask = slots - 1;
            hipLaunchKernelGGL(k_fill_u64, dim3(2048), dim3(256), 0, c->stream, (uint64_t*)c->set.p, slots * 2, ~0ull);
            for (const dfk_ctx::Part& pt : c->parts)
                if (pt.n_blist)
                    hipLaunchKernelGGL(k_set_insert_list, dim3((unsigned)((pt.n_blist + 255) / 256)), dim3(256), 0, c->stream, (const uint4*)pt.buf.p,
                                       (const uint32_t*)((const char*)pt.buf.p + pt.blist + 16), pt.n_blist, (SetSlot*)c->set.p, c->set_mask);
            HIP_TRY(hipGetLastError());
        }
        DevBuf d_n; rc = c->alloc(d_n, 16, "probe counter"); if (rc) return rc;
        HIP_TRY(hipMemsetAsync(d_n.p, 0, 16, c->stream));
        for (const dfk_ctx::Part& pt : c->parts) {
            if (!pt.n) continue;
            if (listed) {
                if (pt.n_blist)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_adjacency_list<K>), dim3((unsigned)std::min<uint64_t>((pt.n_blist + 255) / 256, 16384)), dim3(256), 0,
                                       c->stream, (uint4*)pt.buf.p, (const uint32_t*)((const char*)pt.buf.p + pt.blist + 16), pt.n_blist,
                                       (const SetSlot*)c->set.p, c->set_mask, (unsigned long long*)d_n.p);
            } else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_adjacency<K>), dim3((unsigned)std::min<uint64_t>((pt.n + 255) / 256, 8192)), dim3(256), 0,
                                   c->stream, (uint4*)pt.buf.p, pt.n, (const SetSlot*)c->set.p, c->set_mask, (unsigned long long*)d_n.p);
        }
        HIP_TRY(hipGetLastError());
        uint64_t np = 0;
        HIP_TRY(hipMemcpyAsync(&np, d_n.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->st.adj_probes = np;
        c->release(d_n); c->release(c->set);
    }
    c->st.ms_adjacency = t.stop();
    TRACE("adjacency done");
    return 0;
}

uint32_t pick_log2_nb(uint64_t n_inst, uint32_t log2_world)
{
    // fine buckets of ~430-850 instances (the count is a power of two); an item packs several of them up to its
    // instance budget.  Measured on configs[1] at K = 40/48/60: 2^27 buckets beat 2^28 (the scan's atomics run on a
    // 1 GB counter table instead of 2 GB: -7..-23 ms) and 2^26 (items overflow their tables: +190 ms).
    // (When sharded, the record header keeps 24 bits of the bucket id inside the pass for the receiver's
    // regroup: dfk_shard_plan asks for enough passes that a pass has <= 2^24 buckets per owner.)
    static const uint64_t per = getenv("DFK_INST_PER_BUCKET") ? (uint64_t)atoll(getenv("DFK_INST_PER_BUCKET")) : 850;
    uint32_t l = ceil_log2(n_inst / per + 1);
    l = std::max<uint32_t>(l, 4 + log2_world);
    return std::min<uint32_t>(l, 28);
}

// How many fine buckets the next pass may take, from what is free now.  A pass holds its bucket tables, its
// records (32 B each) and its dense part of the dictionary (reserved when the pass is counted); the parts of
// earlier passes stay resident, so later passes are smaller.  Buckets are hash-distributed, so a range holds
// its share of the records and instances to within a fraction of a percent.
// `running` != null: the range is scattered while that pass is being counted, so (1) its tables and records
// must fit beside everything the running pass holds, and (2) its part must fit once the running pass has been
// released.
struct RunningPass { uint64_t bytes_held; uint64_t n_inst; uint32_t n_buckets; uint64_t block_off; };

// The same question asked of the arena's actual free blocks (one chunk).  A pass block goes to the highest free
// block that holds it -- now, beside the running pass.  The reservation for its part is made later, when the
// running pass's block is gone, and goes to the lowest free block (the dictionary grows upwards): it has to fit
// there, or it lands higher up and cuts the room of later blocks.  Largest n for which both hold, by bisection.
double fit_free_blocks(const dfk_ctx* c, double per_block, double per_res, const RunningPass& run)
{
    if (c->chunks.size() != 1 || c->chunks[0].free_list.empty()) return 1e300;
    const std::vector<dfk_ctx::Free>& fl = c->chunks[0].free_list;
    auto feasible = [&](double n) {
        const uint64_t B = (uint64_t)(per_block * n / 0.98) + 1;
        size_t at = fl.size();
        for (size_t i = fl.size(); i-- > 0;) if (fl[i].bytes >= B) { at = i; break; }
        if (at == fl.size()) return false;
        // free list once the running block is gone and the new block is in place (at the upper end of fl[at])
        std::vector<dfk_ctx::Free> h(fl.begin(), fl.end());
        h[at].bytes -= B;
        h.push_back(dfk_ctx::Free{run.block_off, run.bytes_held});
        std::sort(h.begin(), h.end(), [](const dfk_ctx::Free& a, const dfk_ctx::Free& b) { return a.off < b.off; });
        uint64_t low_off = 0, low_bytes = 0; bool have = false;
        for (const dfk_ctx::Free& f : h) {                             // the lowest run of adjacent free blocks
            if (!f.bytes) continue;
            if (!have) { low_off = f.off; low_bytes = f.bytes; have = true; }
            else if (low_off + low_bytes == f.off) low_bytes += f.bytes;
            else break;
        }
        return per_res * n <= 0.98 * (double)low_bytes;
    };
    double lo = 0.0, hi = 0.0;
    for (const dfk_ctx::Free& f : fl) hi = std::max(hi, 0.98 * (double)f.bytes / per_block);
    if (feasible(hi)) return hi;
    for (int it = 0; it < 30; ++it) { const double mid = 0.5 * (lo + hi); if (feasible(mid)) lo = mid; else hi = mid; }
    return lo;
}

uint32_t plan_range(const dfk_ctx* c, const BucketTable& T, const CountRun& R, uint32_t sub_nb, uint32_t lo, const RunningPass* running, bool overlap)
{
    const double room = c->budget > c->held ? (double)(c->budget - c->held) : 0.0;
    const double inst_per = (double)T.n_inst / sub_nb, rec_per = (double)T.n_records / sub_nb;
    // solid k-mers per instance: observed on the passes done so far, else the prior stage_count starts from
    double ratio = R.inst_seen ? 1.12 * (double)R.solid_seen / (double)R.inst_seen : 1.0 / 10.0;
    ratio = std::min(ratio, 1.0 / std::max<uint32_t>(1, c->cfg.min_freq));
    const double per_in = 81.0 + 32.5 * rec_per;                                         // tables and records (as the pass block is sized)
    const double per_seg = 0.0;                                                                       // (workgroups write straight into the part's reservation: nothing else per pass)
    const double per_out = 32.0 * ratio * inst_per * (double)(1u << c->seg_attempt);                  // the part's reservation
    const double fixed = 120e6;                                       // chunk ends left empty by the workgroups (67 MB), small tables
    const double left = (double)(sub_nb - lo);
    double fit;
    if (!running) {
        fit = room > fixed ? (room - fixed) / (per_in + per_out) : 0.0;
        // more than one pass to go: leave room for the records of the next one, which is scattered while this
        // one is counted (passes of equal or decreasing size also keep the arena from fragmenting: each new
        // range fits the hole left by the pass before the running one)
        if (c->plan_derate * fit < left && overlap) {
            fit = room > fixed ? (room - fixed) / (2.0 * per_in + per_seg + per_out) : 0.0;
            // the very first scatter has nothing to hide under: keep it short (its fixed cost, reading every
            // summary, is paid anyway; a tenth of the buckets -- a sixteenth while k_count ran at 75 G instances/s
            // and a range could grow by 1.3 from pass to pass -- measured: DFK_PLAN_FIRST / DFK_PLAN_GROWTH, tools/plan_sweep.sh)
            static const double first_div = getenv("DFK_PLAN_FIRST") ? atof(getenv("DFK_PLAN_FIRST")) : 10.0;
            if (lo == 0) fit = std::min(fit, (double)sub_nb / first_div / c->plan_derate);
        }
    } else {
        const double now = room - fixed;                              // (the running pass's part is reserved already)
        const double later = room + (double)running->bytes_held - fixed;
        TRACE("plan: now %.1f%% later %.1f%% geometry %.1f%% balance %.1f%% (room %.1f GB, running block %.1f GB)",
              100 * 0.9 * now / (per_in + per_seg) / sub_nb, 100 * later / (per_in + per_out) / sub_nb,
              100 * fit_free_blocks(c, per_in + per_seg, per_out - per_seg, *running) / sub_nb,
              100 * std::max(0.0, room + (double)running->bytes_held - fixed) / (2.0 * (per_in + per_seg) + (per_out - per_seg)) / sub_nb,
              room / 1e9, running->bytes_held / 1e9);
        if (g_trace) for (const dfk_ctx::Free& f : c->chunks[0].free_list) fprintf(stderr, "[dfk]   free %.2f GB at %.2f GB\n", f.bytes / 1e9, f.off / 1e9);
        // (a range that does not fit beside the running pass loses the overlap: plan it with more slack --
        // the free room is in several pieces by now)
        fit = std::max(0.0, std::min(0.9 * now / (per_in + per_seg), later / (per_in + per_out)));
        fit = std::min(fit, fit_free_blocks(c, per_in + per_seg, per_out - per_seg, *running) / c->plan_derate);
        // and leave the pass after this one (planned while this one is counted, the running one gone by then)
        // a block of the same size: greedy ranges alternate between huge and tiny
        fit = std::min(fit, std::max(0.0, room + (double)running->bytes_held - fixed) / (2.0 * (per_in + per_seg) + (per_out - per_seg)));
    }
    double n = c->plan_derate * fit;
    // the range is scattered while the running pass is counted: no larger than what that count hides (beside
    // k_count a sweep moves a range's records about 1.15 times as fast as k_count counts them -- 8.3 ms against 9.6 ms
    // per percent of the human-scale set's buckets -- after ~10 ms of reading masks; with 1.3, the value from when
    // k_count ran at 75 G instances/s, the first three counts each waited 13-18 ms for the records of the next)
    static const double growth = getenv("DFK_PLAN_GROWTH") ? atof(getenv("DFK_PLAN_GROWTH")) : 1.1;
    if (running && overlap) n = std::min(n, growth * (double)running->n_buckets);
    if (left <= 0.99 * fit && left < 1.06 * n) n = left;              // no sliver of a last pass if the rest (almost certainly) fits
    else if (left > n && left < 1.3 * n) n = 0.5 * left + 1.0;        // two even passes rather than a big one and a sliver (1.7 while the cliff of section 9 was unexplained: a big count beside a small sweep)
    n = std::min(n, left);
    if (n < 16.0) return running ? 0u : (uint32_t)std::min(16.0, left);
    return (uint32_t)n;
}

struct StreamSwap {                                    // run a stretch of host code against the second stream
    dfk_ctx* c; hipStream_t keep;
    StreamSwap(dfk_ctx* ctx, hipStream_t s) : c(ctx), keep(ctx->stream) { c->stream = s; }
    ~StreamSwap() { c->stream = keep; }
};

// trim and counting scan done ahead of run_typed, while the reads' bases were still crossing PCIe (count_host)
struct Prescan { BucketTable T; uint64_t n_inst = 0; float ms_trim = 0; };

template <int K>
int run_typed(dfk_ctx* c, const Inputs& in, Prescan* pre = nullptr)
{
    Timer total(c->stream);
    total.start();
    Timer t(c->stream);
    uint64_t n_inst = 0;
    int rc = 0;
    BucketTable T;
    if (pre) { n_inst = pre->n_inst; c->st.ms_trim = pre->ms_trim; T = std::move(pre->T); }
    else {
        t.start();
        TRACE("trim: %llu reads", (unsigned long long)in.n_reads);
        rc = stage_trim<K>(c, in, &n_inst); if (rc) return rc;
        c->st.ms_trim = t.stop();
        TRACE("trim done: %llu instances", (unsigned long long)n_inst);
    }
    c->st.n_reads = in.n_reads; c->st.n_inst = n_inst; c->n_reads = in.n_reads;
    if (!pre) {
        T.log2_nb = pick_log2_nb(n_inst, 0);
        rc = partition_count<K>(c, in, n_inst, 0, 0, &T); if (rc) return rc;
    }
    CountRun R;
    rc = count_run_begin(c, &R); if (rc) return rc;
    c->want_blist = true; c->pending_blist = -1;
    // Passes over contiguous ranges of the fine buckets, each as large as the free HBM allows.  While a pass is
    // counted (LDS- and issue-bound) the next range is scattered on a second, low-priority stream (bound by
    // scattered atomics and stores): the two kernels share the CUs.  A pass that runs out of room (its estimate
    // of the solid k-mers was too low) is undone together with whatever was started after it, and redone smaller.
    const uint32_t sub_nb = 1u << T.log2_nb;
    const uint32_t forced = (uint32_t)c->cfg.reserved[0];            // dfk_config.reserved[0] = forced number of passes (tests)
    const uint32_t per_forced = forced ? std::max<uint32_t>(1, (sub_nb + forced - 1) / forced) : 0;
    const bool overlap = getenv("DFK_NO_OVERLAP") == nullptr;
    uint32_t n_passes = 0, retries = 0;
    c->seg_attempt = 0; c->distinct_per_inst = 0.0;
    struct Job { ScatterJob sj; dfk_ctx::PassBlock blk; uint64_t mark = 0; bool valid = false; };
    Job cur, nxt;
    auto drop_events = [&](Job& j) { if (j.sj.e0) { (void)hipEventDestroy(j.sj.e0); (void)hipEventDestroy(j.sj.e1); j.sj.e0 = j.sj.e1 = nullptr; } };
    // start the scatter of [lo, lo + n) on the second stream; NOMEM leaves nothing behind
    // A range that is scattered beside a running count is launched BEHIND that count's k_count, into what k_count
    // leaves free.  The other order is the 2.5x cliff of DESIGN.md section 9: k_count launched into a chip on which
    // short-lived sweep blocks (4 KB of LDS each) are already resident gets its 71-KB workgroups placed between them,
    // and a workgroup that lands in the middle of a CU's 160 KB leaves no contiguous 71 KB for the second one as long
    // as it lives -- it is persistent, so for the whole launch: half the workgroups on that CU, also after the sweep
    // has ended (timeline: profiles/r02_cliff_timeline.txt; with the kernels serialised under --pmc every pass runs at
    // the same 0.254 cycles per instance).  It was first seen on small ranges, whose predecessor's sweep was still
    // running; with the sweep launched microseconds ahead of k_count it is a race that large ranges lose too, now
    // and then (one step in four under a kernel trace, whose launch overhead gives the sweep a head start: pass 3 at
    // 272 ms instead of 116).  Behind k_count a sweep runs a few percent longer: 4 ms per step (tools/defer_test.sh).
    static const double defer_below = getenv("DFK_DEFER_SWEEP_BELOW") ? atof(getenv("DFK_DEFER_SWEEP_BELOW")) : 1.0;
    auto start = [&](Job& j, uint32_t lo, uint32_t n, bool may_defer = false) -> int {
        j.sj = ScatterJob{}; j.mark = c->alloc_seq; j.valid = false;
        const bool defer = may_defer && (double)n < defer_below * (double)sub_nb;
        TRACE("pass range [%u, %u) of %u (%.1f %%), %.2f GB held of %.2f", lo, lo + n, sub_nb, 100.0 * n / sub_nb, c->held / 1e9, c->budget / 1e9);
        // the pass's block: tables (80 B per bucket with their scratch) and records (an estimate: what does not
        // fit the block falls back to the open arena)
        const double share = (double)n / sub_nb;
        const uint64_t blk_bytes = (uint64_t)(1.01 * (80.0 * n + 32.0 * share * (double)T.n_records)) + (16ull << 20);
        j.blk = dfk_ctx::PassBlock{};
        int r = c->alloc(j.blk.block, blk_bytes, "pass block");
        if (!r) {
            c->sub = &j.blk;
            { StreamSwap sw(c, c->stream2); r = scatter_begin<K>(c, in, T, 0, 0, lo, n, &j.sj, !defer); }
            c->sub = nullptr;
        }
        if (r) { (void)hipStreamSynchronize(c->stream2); c->release_since(j.mark); drop_events(j); return r; }
        if (defer) { ScatterJob* sj = &j.sj; c->after_count_launch = [c, &in, &T, sj]() { return scatter_launch<K>(c, in, T, 0, 0, sj); }; TRACE("  (its sweep starts behind k_count)"); }
        j.valid = true;
        return 0;
    };
    // the first range of a (re)started sequence: alone, shrinking until it fits
    auto start_alone = [&](Job& j, uint32_t lo) -> int {
        for (;;) {
            const uint32_t n = forced ? std::min(per_forced, sub_nb - lo) : plan_range(c, T, R, sub_nb, lo, nullptr, overlap);
            // a budget that leaves room for slivers only would take thousands of passes: say so instead
            if (!forced && n < sub_nb - lo && (uint64_t)n * 1024 < sub_nb)
                return fail(DFK_E_NOMEM, "HBM budget too small: %.2f GB free of %.2f GB allows passes of %u of %u buckets",
                            (c->budget - c->held) / 1e9, c->budget / 1e9, n, sub_nb);
            const int r = start(j, lo, n);
            if (r != DFK_E_NOMEM || forced || n <= 16 || ++retries > 12) return r;
            c->plan_derate *= 0.7;
            TRACE("out of HBM (%s): planning smaller passes", g_err.c_str());
        }
    };
    rc = start_alone(cur, 0); if (rc) return rc;
    while (cur.valid) {
        const uint32_t lo = cur.sj.lo, n = cur.sj.n, nlo = lo + n;
        rc = scatter_end<K>(c, in, T, 0, 0, &cur.sj); if (rc) return rc;
        const dfk_stats st0 = c->st;
        rc = count_snapshot(c, &R, false); if (rc) return rc;
        c->sub = &cur.blk;
        rc = count_prepare<K>(c, cur.sj.P, R, barcode_words(c, in.bc != nullptr));
        c->sub = nullptr;
        nxt.valid = false;
        if (!rc && overlap && nlo < sub_nb) {
            const Partition& P = cur.sj.P;
            (void)P;
            const RunningPass rp{cur.blk.block.bytes, cur.sj.P.n_inst, n, (uint64_t)((char*)cur.blk.block.p - c->chunks[0].p)};
            const uint32_t n2 = forced ? std::min(per_forced, sub_nb - nlo) : plan_range(c, T, R, sub_nb, nlo, &rp, overlap);
            if (n2) {
                const int r2 = start(nxt, nlo, n2, true);
                if (r2 && r2 != DFK_E_NOMEM) return r2;                // NOMEM: this range is scattered after the count instead
            }
        }
        if (!rc) {
            switch (barcode_words(c, in.bc != nullptr)) {               // barcodes a table slot remembers: max(1, MIN_BC - 1)
            case 0: rc = count_run<K, 0>(c, cur.sj.P, R); break;
            case 1: rc = count_run<K, 1>(c, cur.sj.P, R); break;
            case 2: rc = count_run<K, 2>(c, cur.sj.P, R); break;
            case 3: rc = count_run<K, 3>(c, cur.sj.P, R); break;
            case 4: rc = count_run<K, 4>(c, cur.sj.P, R); break;
            case 5: rc = count_run<K, 5>(c, cur.sj.P, R); break;
            case 6: rc = count_run<K, 6>(c, cur.sj.P, R); break;
            default: rc = count_run<K, 7>(c, cur.sj.P, R); break;
            }
        }
        if (c->after_count_launch) {                                     // (no k_count was launched: an error on the way there)
            std::function<int()> f; f.swap(c->after_count_launch);
            if (!rc) { const int r3 = f(); if (r3) return r3; } else nxt.valid = false;
        }
        if (rc == DFK_E_NOMEM || rc == E_SEGMENT_FULL) {
            HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipStreamSynchronize(c->stream2));
            c->release_since(cur.mark);                               // this pass and the one started under it
            R.d_wg = R.big = R.d_part = DevBuf{};
            drop_events(cur); drop_events(nxt); nxt.valid = false;
            int rc2 = count_snapshot(c, &R, true); if (rc2) return rc2;
            const float ms_scatter = c->st.ms_part_scatter, ms_count = c->st.ms_count, ms_fb = c->st.ms_fallback;
            c->st = st0;                                              // the time spent stays on the books
            c->st.ms_part_scatter = ms_scatter; c->st.ms_count = ms_count; c->st.ms_fallback = ms_fb;
            if (++retries > 12) return DFK_E_NOMEM;
            if (rc == E_SEGMENT_FULL) { ++c->seg_attempt; TRACE("the room reserved for the pass's solid k-mers was too small: redoing the pass with twice as much"); }
            else if (forced || n <= 16) return rc;
            else { c->plan_derate *= 0.7; TRACE("out of HBM (%s): redoing the pass smaller", g_err.c_str()); }
            rc = start_alone(cur, lo); if (rc) return rc;
            continue;
        }
        if (rc) return rc;
        release_pass(c, &cur.sj.P);
        c->release(cur.blk.block);
        drop_events(cur);
        ++n_passes;
        c->seg_attempt = 0;                                           // later passes size their output from the observed ratio
        if (nxt.valid) { cur = nxt; nxt = Job{}; }
        else { cur = Job{}; if (nlo < sub_nb) { rc = start_alone(cur, nlo); if (rc) return rc; } }
    }
    c->release(T.acc); c->release(T.summ); c->release(T.classes); c->release(T.ovf_list); c->release(T.class_hist);
    c->st.reserved[0] = n_passes;
    rc = count_run_end(c, &R); if (rc) return rc;
    rc = stage_adjacency<K>(c); if (rc) return rc;
    c->st.ms_total = total.stop() + (pre ? c->st.ms_trim + c->st.ms_part_count : 0.0f);   // (device time: what ran under the upload counts too)
    c->st.hbm_bytes_peak = c->peak;
    c->have = true;
    return 0;
}

int run(dfk_ctx* c, const Inputs& in)
{
    int rc;
    switch (c->cfg.K) {
    case 40: rc = run_typed<40>(c, in); break;
    case 48: rc = run_typed<48>(c, in); break;
    case 60: rc = run_typed<60>(c, in); break;
    default: return fail(DFK_E_ARG, "K must be 40, 48 or 60");
    }
    return rc == E_SEGMENT_FULL ? DFK_E_NOMEM : rc;
}

int upload(dfk_ctx* c, void* d, const void* h, uint64_t bytes);

// count_host's way through a run: everything but the bases is on the device (in.packed is allocated and empty), the bases
// are at `h_packed` on the host.  The trim needs no bases and runs beside the first piece of their upload; the counting scan
// -- a fifth of the count, serial, nothing else can start before it -- takes the reads of every piece as soon as the piece
// has arrived, so that only the last piece's scan is left when the upload ends (the reads a piece completes: from every
// stride-th entry of the offset table, read back once).  Then the passes as in any run.
template <int K>
int run_under_upload(dfk_ctx* c, const Inputs& in, const uint8_t* h_packed)
{
    const uint64_t n = in.n_reads, pb = in.packed_bytes;
    const uint64_t seg_env = getenv("DFK_UPLOAD_SEGMENT") ? (uint64_t)atoll(getenv("DFK_UPLOAD_SEGMENT")) : 0;      // (tests: small pieces)
    const uint64_t seg = seg_env ? std::max<uint64_t>(64, seg_env & ~63ull) : std::max<uint64_t>(256ull << 20, ((pb / 12) + (64ull << 20)) & ~((64ull << 20) - 1));
    // where the reads end, coarsely
    const uint64_t stride = std::max<uint64_t>(1, (n + 32767) / 32768), n_samp = n / stride + 2;
    std::vector<uint64_t> samp(n_samp);
    {
        DevBuf d; int rc = c->alloc(d, n_samp * 8, "offset samples", true); if (rc) return rc;
        hipLaunchKernelGGL(k_sample_u64, dim3((unsigned)((n_samp + 255) / 256)), dim3(256), 0, c->stream, in.base_off, stride, n, n_samp, (uint64_t*)d.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(samp.data(), d.p, n_samp * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->release(d);
    }
    auto reads_within = [&](uint64_t bytes) -> uint64_t {              // reads whose bases end at or before `bytes` (a multiple of the stride, or all)
        const uint64_t i = (uint64_t)(std::upper_bound(samp.begin(), samp.end(), bytes) - samp.begin());   // samp[i-1] <= bytes < samp[i]
        return i ? std::min<uint64_t>((i - 1) * stride, n) : 0;
    };
    Prescan pre;
    // ---- the trim beside the first piece
    int trim_rc = 0; std::string trim_err;
    Timer tt(c->stream);
    std::thread trim([&] {
        (void)hipSetDevice(c->device);
        TRACE("trim: %llu reads (the bases are still arriving)", (unsigned long long)n);
        tt.start();
        trim_rc = stage_trim<K>(c, in, &pre.n_inst);
        if (trim_rc) trim_err = g_err; else pre.ms_trim = tt.stop();
    });
    struct JoinTrim { std::thread& t; ~JoinTrim() { if (t.joinable()) t.join(); } } join_trim{trim};     // (whatever happens below)
    uint64_t sent = std::min(seg, pb);
    int rc = upload(c, (void*)in.packed, h_packed, sent);
    if (sent >= pb) c->t_upload_done = wall_now();
    trim.join();
    if (trim_rc) return fail(trim_rc, "%s", trim_err.c_str());
    if (rc) return rc;
    TRACE("trim done: %llu instances", (unsigned long long)pre.n_inst);
    // ---- the scan, a piece behind the upload
    pre.T.log2_nb = pick_log2_nb(pre.n_inst, 0);
    ScanJob J;
    rc = scan_begin<K>(c, in, 0, 0, &pre.T, false, &J); if (rc) return rc;
    uint64_t scanned = 0;
    for (;;) {
        const uint64_t r1 = sent >= pb ? n : reads_within(sent);
        rc = scan_range<K>(c, in, &pre.T, &J, scanned, r1); if (rc) return rc;
        scanned = std::max(scanned, r1);
        if (sent >= pb) break;
        const uint64_t m = std::min(seg, pb - sent);
        rc = upload(c, (char*)in.packed + sent, h_packed + sent, m); if (rc) return rc;
        sent += m;
        if (sent >= pb) c->t_upload_done = wall_now();
    }
    rc = scan_end<K>(c, in, pre.n_inst, &pre.T, &J); if (rc) return rc;
    return run_typed<K>(c, in, &pre);
}

int run_from_host_bases(dfk_ctx* c, const Inputs& in, const uint8_t* h_packed)
{
    int rc;
    switch (c->cfg.K) {
    case 40: rc = run_under_upload<40>(c, in, h_packed); break;
    case 48: rc = run_under_upload<48>(c, in, h_packed); break;
    case 60: rc = run_under_upload<60>(c, in, h_packed); break;
    default: return fail(DFK_E_ARG, "K must be 40, 48 or 60");
    }
    return rc == E_SEGMENT_FULL ? DFK_E_NOMEM : rc;
}
bool scan_under_upload_applies(const dfk_ctx* c, uint64_t packed_bytes)
{
    // (read at every call: tests switch them within one process)
    const bool off = getenv("DFK_NO_SCAN_UNDER_UPLOAD") != nullptr;
    const uint64_t min_bytes = getenv("DFK_SCAN_UNDER_UPLOAD_MIN") ? (uint64_t)atoll(getenv("DFK_SCAN_UNDER_UPLOAD_MIN")) : (1ull << 30);
    if (off || packed_bytes < min_bytes) return false;
    switch (c->cfg.K) { case 40: return scan_takes_ranges<40>(c); case 48: return scan_takes_ranges<48>(c); case 60: return scan_takes_ranges<60>(c); }
    return false;
}

// Ascending (w0, w1) on the host cores: one most-significant-digit pass (12 bits of w0) over per-thread slices,
// then the 4096 key ranges are sorted independently by a pool of threads.  A human-scale dictionary is
// 3x10^9 entries; a single std::sort of that takes many minutes.
void host_sort_entries(std::vector<dfk_entry32>& v)
{
    const auto less = [](const dfk_entry32& a, const dfk_entry32& b) { return a.w0 != b.w0 ? a.w0 < b.w0 : a.w1 < b.w1; };
    const uint64_t n = v.size();
    const unsigned T = (unsigned)std::min<uint64_t>(std::max(1u, std::thread::hardware_concurrency()), std::max<uint64_t>(1, n >> 14));
    if (T <= 1) { std::sort(v.begin(), v.end(), less); return; }
    constexpr unsigned B = 4096, SH = 52;
    std::vector<dfk_entry32> tmp(n);
    std::vector<uint64_t> cnt((size_t)T * B, 0);
    auto slice = [&](unsigned t) { return std::pair<uint64_t, uint64_t>(n * t / T, n * (t + 1) / T); };
    auto run = [&](auto&& body) { std::vector<std::thread> th; for (unsigned t = 0; t < T; ++t) th.emplace_back(body, t); for (auto& x : th) x.join(); };
    run([&](unsigned t) { auto [a, b] = slice(t); uint64_t* c = &cnt[(size_t)t * B]; for (uint64_t i = a; i < b; ++i) ++c[v[i].w0 >> SH]; });
    std::vector<uint64_t> start(B + 1, 0);
    {   // bucket-major, thread-minor offsets
        uint64_t at = 0;
        for (unsigned b = 0; b < B; ++b) { start[b] = at; for (unsigned t = 0; t < T; ++t) { const uint64_t c = cnt[(size_t)t * B + b]; cnt[(size_t)t * B + b] = at; at += c; } }
        start[B] = at;
    }
    run([&](unsigned t) { auto [a, b] = slice(t); uint64_t* c = &cnt[(size_t)t * B]; for (uint64_t i = a; i < b; ++i) tmp[c[v[i].w0 >> SH]++] = v[i]; });
    std::atomic<unsigned> next{0};
    run([&](unsigned) { for (unsigned b; (b = next.fetch_add(1)) < B;) std::sort(tmp.begin() + start[b], tmp.begin() + start[b + 1], less); });
    v.swap(tmp);
}

// ------------------------------------------------------------------ host <-> device transfers at scale
// A 30x human run uploads ~100 GB of reads and hands back a 99 GB dictionary.  One pageable hipMemcpy moves that at
// a few GB/s (the runtime stages it through one bounce buffer on one core); here a few host threads each own two
// pinned buffers and a stream, pull chunks off a shared counter, and overlap their memcpy (or file write) with the
// DMA of their other buffer.  Small transfers take the plain call.
// (DFK_XFER_CHUNK: the tests push a few hundred reads through the many-chunk path)
// (4 MiB: a 45-GB upload takes 1.65 s; 2.3 s with 16-MiB pieces, 3.1 s with 32, 1.8 s with 1-2 MiB, and at 512 KiB the
// dictionary's way out slows from 7.6 to 9 s -- tools/xfer_sweep.sh)
const size_t XFER_CHUNK = [] { const char* e = getenv("DFK_XFER_CHUNK"); return e && atoll(e) >= 64 ? ((size_t)atoll(e) & ~(size_t)31) : (size_t)4 << 20; }();
unsigned xfer_threads()
{
    static const unsigned n = [] {
        if (const char* e = getenv("DFK_HOST_THREADS")) return (unsigned)std::max(1, atoi(e));
        return std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    }();
    return n;
}

struct XferLane { void* pin[2] = {nullptr, nullptr}; hipStream_t st = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; unsigned turn = 0; };

// body(t, lane, i) is called for chunk indices i = 0..n_chunks-1, each exactly once, from T = min(n_chunks,
// xfer_threads()) threads bound to the context's device (t = thread number, lane = its buffers and stream);
// fin(t, lane) once per thread when the chunks have run out.  The first non-zero return stops the rest.
using XferBody = std::function<int(unsigned, XferLane&, uint64_t)>;
using XferFin = std::function<int(unsigned, XferLane&)>;
struct LanePool { std::vector<XferLane> idle; };
void lane_destroy(XferLane& l)
{
    for (int k = 0; k < 2; ++k) { if (l.pin[k]) (void)hipHostFree(l.pin[k]); if (l.ev[k]) (void)hipEventDestroy(l.ev[k]); }
    if (l.st) (void)hipStreamDestroy(l.st);
    l = XferLane{};
}
void lane_pool_delete(void* p) { LanePool* P = (LanePool*)p; for (XferLane& l : P->idle) lane_destroy(l); delete P; }
int lane_create(XferLane& l)
{
    for (int k = 0; k < 2; ++k) {
        if (hipHostMalloc(&l.pin[k], XFER_CHUNK, hipHostMallocDefault) != hipSuccess) return fail(DFK_E_NOMEM, "cannot pin %zu bytes of host memory for transfers", XFER_CHUNK);
        if (hipEventCreateWithFlags(&l.ev[k], hipEventDisableTiming) != hipSuccess) return fail(DFK_E_HIP, "hipEventCreate failed");
    }
    if (hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking) != hipSuccess) return fail(DFK_E_HIP, "hipStreamCreate failed");
    return 0;
}

int xfer_run(dfk_ctx* c, uint64_t n_chunks, const XferBody& body, const XferFin& fin = nullptr, unsigned n_lanes = 0)
{
    const unsigned T = (unsigned)std::min<uint64_t>(n_chunks, n_lanes ? n_lanes : xfer_threads());
    if (!T) return 0;
    // lanes: from the context's pool (several transfers may run side by side, each with lanes of its own), made when it has none
    std::vector<XferLane> lanes(T);
    int rc = 0;
    {
        std::lock_guard<std::mutex> g(c->lane_mu);
        if (!c->lane_pool) { c->lane_pool = new LanePool; c->lane_pool_free = lane_pool_delete; }
        LanePool* P = (LanePool*)c->lane_pool;
        for (XferLane& l : lanes) if (!P->idle.empty()) { l = P->idle.back(); P->idle.pop_back(); }
    }
    for (XferLane& l : lanes) if (!rc && !l.st) rc = lane_create(l);
    std::atomic<uint64_t> next{0};
    std::atomic<int> err{0};
    std::string err_msg;
    std::mutex mu;
    if (!rc) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                (void)hipSetDevice(c->device);
                auto note = [&](int r) { if (r) { std::lock_guard<std::mutex> g(mu); if (!err.load()) { err_msg = g_err; err = r; } } };
                for (uint64_t i; !err.load() && (i = next.fetch_add(1)) < n_chunks;) note(body(t, lanes[t], i));
                if (fin && !err.load()) note(fin(t, lanes[t]));
                (void)hipStreamSynchronize(lanes[t].st);
            });
        for (std::thread& x : th) x.join();
        if (err.load()) { rc = err.load(); g_err = err_msg; }
    }
    {   // idle again (every lane's stream has been waited for): back to the pool, which keeps a few
        std::lock_guard<std::mutex> g(c->lane_mu);
        LanePool* P = (LanePool*)c->lane_pool;
        for (XferLane& l : lanes) {
            if (!l.st || !l.pin[0] || !l.pin[1] || !l.ev[0] || !l.ev[1] || P->idle.size() >= 24) { lane_destroy(l); continue; }
            l.turn = 0;
            P->idle.push_back(l);
        }
    }
    return rc;
}

// `bytes` of the caller's host memory at h.  Inside a range it has hinted (dfk_hint_file_range): of the file behind it, if a
// descriptor came with the hint; else out of the memory, whose pages are then dropped from the caller's page table (the file
// keeps them) -- by the lane that copied them, so that no single thread is left to unmap 90 GB of touched pages.
int host_read(dfk_ctx* c, void* dst, const void* h, uint64_t bytes)
{
    for (const dfk_ctx::FileRange& r : c->file_ranges)
        if ((const char*)h >= r.base && (const char*)h + bytes <= r.base + r.bytes) {
            if (r.fd < 0) {
                memcpy(dst, h, bytes);
                const uintptr_t page = 4096, lo = ((uintptr_t)h + page - 1) & ~(page - 1), hi = ((uintptr_t)h + bytes) & ~(page - 1);
                if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_DONTNEED);
                return 0;
            }
            uint64_t at = r.off + (uint64_t)((const char*)h - r.base);
            for (uint64_t done = 0; done < bytes;) {
                const ssize_t got = pread(r.fd, (char*)dst + done, bytes - done, (off_t)(at + done));
                if (got <= 0) return fail(DFK_E_INPUT, "cannot read %llu bytes at %llu of the file behind a hinted input range", (unsigned long long)(bytes - done), (unsigned long long)(at + done));
                done += (uint64_t)got;
            }
            return 0;
        }
    memcpy(dst, h, bytes);
    return 0;
}

// host -> device, `bytes` from pageable (or mapped-file) memory
int upload(dfk_ctx* c, void* d, const void* h, uint64_t bytes)
{
    if (!bytes) return 0;
    if (bytes < 4 * XFER_CHUNK) {
        std::vector<char> tmp(bytes);
        int rc = host_read(c, tmp.data(), h, bytes); if (rc) return rc;
        HIP_TRY(hipMemcpy(d, tmp.data(), bytes, hipMemcpyHostToDevice));
        return 0;
    }
    // (uploads run alone -- the writers of the stage's last phase run three at a time -- and are bound by the lanes' copies out of
    // the page cache, not by PCIe: they may take more lanes than xfer_threads(); DFK_UPLOAD_THREADS)
    static const unsigned up_lanes = getenv("DFK_UPLOAD_THREADS") ? (unsigned)std::max(1, atoi(getenv("DFK_UPLOAD_THREADS"))) : 0;
    return xfer_run(c, (bytes + XFER_CHUNK - 1) / XFER_CHUNK, [&](unsigned, XferLane& l, uint64_t i) -> int {
        const int k = l.turn++ & 1;                                  // the lane's two buffers alternate; the one about to be
        HIP_TRY(hipEventSynchronize(l.ev[k]));                       // overwritten must have left the host
        const uint64_t off = i * XFER_CHUNK, n = std::min<uint64_t>(XFER_CHUNK, bytes - off);
        { const int rr = host_read(c, l.pin[k], (const char*)h + off, n); if (rr) return rr; }
        HIP_TRY(hipMemcpyAsync((char*)d + off, l.pin[k], n, hipMemcpyHostToDevice, l.st));
        HIP_TRY(hipEventRecord(l.ev[k], l.st));
        return 0;
    }, nullptr, up_lanes);
}

// The dictionary as the device holds it (pass after pass, no order inside a pass), streamed into a kmers.kvec image:
// every chunk has its place in the file, so the lanes write independently (pwrite).
int write_parts_unsorted(dfk_ctx* c, int fd, bool pre, uint64_t first_byte = 16)
{
    struct Piece { const char* src; uint64_t bytes, file_off; };
    std::vector<Piece> pieces;
    uint64_t at = first_byte;
    for (const dfk_ctx::Part& pt : c->parts) {
        const DevBuf& src = pre ? pt.pre : pt.buf;
        for (uint64_t o = 0; o < pt.n * 32; o += XFER_CHUNK) pieces.push_back(Piece{(const char*)src.p + o, std::min<uint64_t>(XFER_CHUNK, pt.n * 32 - o), at + o});
        at += pt.n * 32;
    }
    struct Pending { uint64_t bytes = 0, file_off = 0; bool live = false; };
    std::vector<Pending> pend(2 * (size_t)xfer_threads());
    // The lanes store through a shared mapping of the file: concurrent pwrite()s to ONE file serialise on its inode
    // lock (measured on tmpfs: 3.7 GB/s with 16 lanes, 50 GB in 13 s), page faults on a mapping do not.
    // Measured on the GPU box's tmpfs (tools/fs_write_scaling.cc, 16 GB into one file): ONE thread pwrite()s 8.6 GB/s,
    // 16 threads 6.6 GB/s (they serialise on the file's page-cache lock), a shared mapping 3.7-5.2 GB/s.  So the lanes
    // keep the device copies in flight in parallel and take turns at the file: one writer at a time.
    char* map = nullptr;
    if (getenv("DFK_KVEC_MMAP") && at > 16) { map = (char*)mmap(nullptr, at, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (map == (char*)MAP_FAILED) map = nullptr; }
    std::mutex file_turn;
    auto flush = [&](XferLane& l, unsigned t, int k) -> int {      // the chunk sitting in buffer k goes to the file
        Pending& p = pend[2 * t + k];
        if (!p.live) return 0;
        HIP_TRY(hipEventSynchronize(l.ev[k]));
        if (map) memcpy(map + p.file_off, l.pin[k], p.bytes);
        else {
            std::lock_guard<std::mutex> turn(file_turn);
            for (uint64_t done = 0; done < p.bytes;) {
                const ssize_t w = pwrite(fd, (const char*)l.pin[k] + done, p.bytes - done, (off_t)(p.file_off + done));
                if (w <= 0) return fail(DFK_E_ARG, "short write to the k-mer file");
                done += (uint64_t)w;
            }
        }
        p.live = false;
        return 0;
    };
    const int rc = xfer_run(c, pieces.size(),
        [&](unsigned t, XferLane& l, uint64_t i) -> int {
            const int k = l.turn++ & 1;
            int r = flush(l, t, k); if (r) return r;
            const Piece& pc = pieces[i];
            HIP_TRY(hipMemcpyAsync(l.pin[k], pc.src, pc.bytes, hipMemcpyDeviceToHost, l.st));
            HIP_TRY(hipEventRecord(l.ev[k], l.st));
            pend[2 * t + k] = Pending{pc.bytes, pc.file_off, true};
            return flush(l, t, k ^ 1);                              // the other buffer's chunk is written while this one is in flight
        },
        [&](unsigned t, XferLane& l) -> int { int r = flush(l, t, 0); return r ? r : flush(l, t, 1); });
    if (map && munmap(map, at) != 0 && !rc) return fail(DFK_E_ARG, "cannot unmap the k-mer file");
    return rc;
}

int fetch_sorted(dfk_ctx* c, bool pre, std::vector<dfk_entry32>** out)
{
    std::vector<dfk_entry32>& v = pre ? c->sorted_pre : c->sorted;
    bool& ok = pre ? c->sorted_pre_ok : c->sorted_ok;
    if (!ok) {
        if (pre && !(c->cfg.flags & DFK_F_KEEP_PRE_ADJ)) return fail(DFK_E_STATE, "pre-adjacency view needs DFK_F_KEEP_PRE_ADJ");
        v.resize(c->n_solid);
        uint64_t at = 0;
        for (const dfk_ctx::Part& pt : c->parts) {
            const DevBuf& src = pre ? pt.pre : pt.buf;
            if (pt.n) HIP_TRY(hipMemcpy(v.data() + at, src.p, pt.n * 32, hipMemcpyDeviceToHost));
            at += pt.n;
        }
        host_sort_entries(v);
        ok = true;
    }
    *out = &v;
    return 0;
}

} // namespace

// ====================================================================== C ABI
extern "C" {

const char* dfk_last_error(void) { return g_err.c_str(); }
int dfk_abi_version(void) { return DFK_ABI_VERSION; }

int dfk_create(const dfk_config* cfg, dfk_ctx** out)
{
    return guarded([&]() -> int {
    if (!cfg || !out) return fail(DFK_E_ARG, "null argument");
    *out = nullptr;
    if (cfg->abi_version != DFK_ABI_VERSION) return fail(DFK_E_ARG, "dfk_config.abi_version %u != %d", cfg->abi_version, DFK_ABI_VERSION);
    if (cfg->K != 40 && cfg->K != 48 && cfg->K != 60) return fail(DFK_E_ARG, "K=%u: the reference instantiates 40, 48 and 60 only", cfg->K);
    if (cfg->min_bc > DFK_MAX_MIN_BC) return fail(DFK_E_ARG, "MIN_BC=%u: a table slot remembers at most %u distinct barcodes (MIN_BC <= %u)", cfg->min_bc, DFK_MAX_MIN_BC - 1, DFK_MAX_MIN_BC);
    if (cfg->min_freq == 0 || cfg->min_freq > 0xFFFFFFu) return fail(DFK_E_ARG, "MIN_FREQ out of range");
    uint32_t M = cfg->minimizer_len ? cfg->minimizer_len : 16;
    if (M < 8 || M > 16 || M >= cfg->K) return fail(DFK_E_ARG, "minimizer_len must be in 8..16");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(DFK_E_NODEVICE, "no HIP device: libdfk has no CPU path");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(DFK_E_ARG, "device %d out of range (%d devices)", cfg->device, ndev);
    dfk_ctx* c = new dfk_ctx;
    c->cfg = *cfg; c->cfg.minimizer_len = M; c->device = cfg->device;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipGetDeviceProperties(&c->prop, c->device));
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0 && !getenv("DFK_ALLOW_OTHER_ARCH")) {
        std::string a = c->prop.gcnArchName; delete c;
        return fail(DFK_E_NODEVICE, "device is %s; libdfk is built for gfx950 (MI355X) only", a.c_str());
    }
    {
        int lo_pri = 0, hi_pri = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri);      // numerically lowest = highest priority
        HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamDefault, hi_pri));
        HIP_TRY(hipStreamCreateWithPriority(&c->stream2, hipStreamDefault, getenv("DFK_S2_SAME_PRIO") ? hi_pri : lo_pri));
    }
    HIP_TRY(hipMalloc((void**)&c->d_resident, 64));
    HIP_TRY(hipMemset(c->d_resident, 0, 64));
    size_t fr = 0, tot = 0;
    HIP_TRY(hipMemGetInfo(&fr, &tot));
    // The budget is what the arena may reserve: never more than 90 % of what is free now, whatever was asked for (the
    // reference's GRAPHMEM=0.9 of the memory actually there, system/System.cc:1073-1078).  A caller that forwards a
    // host-memory figure -- runall.sh passes MAX_MEM_GB=640 -- gets the device's share, not a plan for room that does
    // not exist.
    const uint64_t cap = (uint64_t)(0.9 * (double)fr);
    c->budget = cfg->hbm_budget_bytes ? std::min<uint64_t>(cfg->hbm_budget_bytes, cap) : cap;
    if (cfg->hbm_budget_bytes > cap)
        TRACE("hbm_budget_bytes %.1f GB is more than 90 %% of the free HBM: clamped to %.1f GB", cfg->hbm_budget_bytes / 1e9, cap / 1e9);
    *out = c;
    return 0;
    });
}

void dfk_destroy(dfk_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    c->release_all();
    c->drop_kept();
    c->drop_pool();
    if (c->lane_pool) { c->lane_pool_free(c->lane_pool); c->lane_pool = nullptr; }
    if (c->d_resident) (void)hipFree(c->d_resident);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    delete c;
}

int dfk_count_device(dfk_ctx* c, const void* d_packed, uint64_t packed_bytes, const void* d_base_off,
                     const void* d_read_len, const void* d_pq, uint64_t pq_nbytes, const void* d_pq_off,
                     const void* d_bc, uint64_t n_reads)
{
    return guarded([&]() -> int {
    if (!c) return fail(DFK_E_ARG, "null context");
    if (n_reads && (!d_packed || !d_base_off || !d_read_len || !d_pq || !d_pq_off)) return fail(DFK_E_ARG, "null input array");
    if (n_reads >= (1ull << 32)) return fail(DFK_E_ARG, "more than 2^32-1 reads in one shard");
    HIP_TRY(hipSetDevice(c->device));
    c->release_all();
    c->st = dfk_stats{}; c->peak = 0;
    if (c->d_resident) HIP_TRY(hipMemset(c->d_resident + 1, 0, 4));
    c->first_chunk_hint = 8 * (packed_bytes + pq_nbytes + 28 * n_reads) + (1ull << 30);
    Inputs in{(const uint8_t*)d_packed, packed_bytes, (const uint64_t*)d_base_off, (const uint32_t*)d_read_len,
              (const uint8_t*)d_pq, pq_nbytes, (const uint64_t*)d_pq_off, (const int32_t*)d_bc, n_reads};
    int rc = run(c, in);
    if (rc) c->release_all();
    return rc;
    });
}

int dfk_hint_file_range(dfk_ctx* c, const void* base, uint64_t bytes, int fd, uint64_t file_off)
{
    if (!c) return fail(DFK_E_ARG, "null context");
    if (!base) { c->file_ranges.clear(); return 0; }
    if (!bytes) return fail(DFK_E_ARG, "a hinted range needs a length");
    if (c->file_ranges.size() >= 8) return fail(DFK_E_ARG, "8 hinted ranges are kept at most");
    c->file_ranges.push_back(dfk_ctx::FileRange{(const char*)base, bytes, fd, file_off});
    if (fd < 0) {
        // VM_SEQ_READ on the caller's mapping: pages that leave the page table (the MADV_DONTNEED behind every copy, the caller's
        // munmap) are then not marked accessed one by one -- on a freshly written tmpfs file that is 25 against 130 GB/s for the
        // reading and seconds for the unmapping (tools/fs_read_after_write.cc)
        const uintptr_t page = 4096, lo = (uintptr_t)base & ~(page - 1), hi = ((uintptr_t)base + bytes + page - 1) & ~(page - 1);
        (void)madvise((void*)lo, hi - lo, MADV_SEQUENTIAL);
    }
    return 0;
}

// dfk_count / dfk_count_bci: the reads from host memory.  base_off == NULL: dense bases, the table derived on the device;
// bci != NULL: the barcode index, expanded on the device (neither then crosses PCIe: 14 + 7 GB of configs[1]'s 95).
static int count_host(dfk_ctx* c, const uint8_t* packed, const uint64_t* base_off, const uint32_t* read_len,
                      const uint8_t* pq, const uint64_t* pq_off, const int32_t* bc, const int64_t* bci, uint64_t n_bci, uint64_t n_reads)
{
    if (!c) return fail(DFK_E_ARG, "null context");
    if (n_reads && (!packed || !read_len || !pq || !pq_off)) return fail(DFK_E_ARG, "null input array");
    if (n_reads >= (1ull << 32)) return fail(DFK_E_ARG, "more than 2^32-1 reads in one shard");
    if (bci && n_bci < 2) return fail(DFK_E_ARG, "a barcode index has at least two entries (bci[0] = 0, bci[last] = the number of reads)");
    HIP_TRY(hipSetDevice(c->device));
    c->release_all();
    c->drop_kept();
    // (the host tables may sit at any address -- e.g. inside a mapped feudal file -- so they are not dereferenced as u64)
    uint64_t pb = 0, qb = 0;
    if (n_reads) { int r1 = base_off ? host_read(c, &pb, (const char*)base_off + 8 * n_reads, 8) : 0; if (!r1) r1 = host_read(c, &qb, (const char*)pq_off + 8 * n_reads, 8); if (r1) return r1; }
    // staging copies live outside the context's run allocations (release_all() at the start of a run)
    void *d_packed = nullptr, *d_boff = nullptr, *d_len = nullptr, *d_pq = nullptr, *d_poff = nullptr, *d_bc = nullptr, *d_bci = nullptr;
    Timer t(c->stream);
    t.start();
    const double t_host0 = wall_now();
    int rc = 0;
    auto room = [&](void** d, size_t bytes) {
        if (rc) return;
        // (64 bytes of slack: the kernels read the 2-bit stream as aligned 32-bit words, up to 3 bytes past its end)
        if (hipMalloc(d, bytes + 64) != hipSuccess) { (void)hipGetLastError(); rc = fail(DFK_E_NOMEM, "no room on the device for %zu bytes of input", bytes); }
    };
    auto up = [&](void** d, const void* h, size_t bytes) {
        room(d, bytes);
        if (rc) return;
        const auto t0 = std::chrono::steady_clock::now();
        rc = upload(c, *d, h, bytes);
        TRACE("uploaded %.2f GB in %.3f s", bytes / 1e9, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
    const unsigned cus = (unsigned)c->prop.multiProcessorCount;
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_reads + 256) / 256, 32ull * cus));
    auto derive = [&]() -> int {                                         // (the kernels of the derived arrays run under the uploads that follow)
        if (!base_off) {
            hipLaunchKernelGGL(k_dense_sizes, dim3(grid), dim3(256), 0, c->stream, (const uint32_t*)d_len, n_reads, (uint64_t*)d_boff);
            HIP_TRY(hipGetLastError());
            int r = device_scan(c, (const uint64_t*)d_boff, (uint64_t*)d_boff, n_reads + 1); if (r) return r;
            HIP_TRY(hipMemcpyAsync(&pb, (const uint64_t*)d_boff + n_reads, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        if (bci) {
            hipLaunchKernelGGL(k_expand_bci, dim3(grid), dim3(256), 0, c->stream, (const int64_t*)d_bci, n_bci, n_reads, (int32_t*)d_bc);
            HIP_TRY(hipGetLastError());
        }
        return 0;
    };
    up(&d_len, read_len, n_reads * 4);
    if (base_off) up(&d_boff, base_off, (n_reads + 1) * 8); else room(&d_boff, (n_reads + 1) * 8);
    if (bci) { up(&d_bci, bci, n_bci * 8); room(&d_bc, n_reads * 4); }
    if (!rc && (!base_off || bci)) { rc = derive(); c->drop_empty_chunks(); }     // (the scan's scratch came from the arena: the run's first chunk is still to be sized)
    // (the bases last: the trim reads the qualities only, and the counting scan follows the bases piece by piece -- run_under_upload)
    const bool under = n_reads && scan_under_upload_applies(c, pb);
    up(&d_pq, pq, qb);
    up(&d_poff, pq_off, (n_reads + 1) * 8);
    if (bc && !bci) up(&d_bc, bc, n_reads * 4);
    if (under) room(&d_packed, pb); else up(&d_packed, packed, pb);
    float ms_up = under ? 0.0f : t.stop();
    if (d_bci) { (void)hipFree(d_bci); d_bci = nullptr; }
    if (!rc) {
        uint64_t staged = pb + qb + (n_reads + 1) * 16 + n_reads * 8 + 6 * 64;
        uint64_t saved = c->budget;
        c->budget = c->budget > staged ? c->budget - staged : 0;
        if (!under) rc = dfk_count_device(c, d_packed, pb, d_boff, d_len, d_pq, qb, d_poff, d_bc, n_reads);
        else {
            // (what dfk_count_device does before its run; the arena is empty: release_all above, the derived arrays' scratch dropped)
            c->st = dfk_stats{}; c->peak = 0;
            if (c->d_resident) HIP_TRY(hipMemset(c->d_resident + 1, 0, 4));
            c->first_chunk_hint = 8 * (pb + qb + 28 * n_reads) + (1ull << 30);
            const Inputs in{(const uint8_t*)d_packed, pb, (const uint64_t*)d_boff, (const uint32_t*)d_len, (const uint8_t*)d_pq, qb, (const uint64_t*)d_poff,
                            (const int32_t*)d_bc, n_reads};
            c->t_upload_done = 0;
            rc = run_from_host_bases(c, in, packed);
            if (rc) { (void)hipDeviceSynchronize(); c->release_all(); }         // (a piece's scan may still be running on what is about to be freed)
            (void)t.stop();
            ms_up = c->t_upload_done > t_host0 ? (float)(1e3 * (c->t_upload_done - t_host0)) : 0.0f;     // (to the arrival of the last base; trim and most of the scan are inside it)
            TRACE("inputs on the device %.3f s after the call, counted %.3f s after it", 1e-3 * ms_up, wall_now() - t_host0);
        }
        c->st.ms_upload = ms_up;
        if (!rc && (c->cfg.flags & DFK_F_KEEP_INPUTS)) {
            // the reads stay on the device for dfk_paths_build, and the arena's budget stays reduced by them
            // (the barcodes are createDict's alone: their room goes back at once)
            void* k[6] = {d_packed, d_boff, d_len, d_pq, d_poff, nullptr};
            memcpy(c->kept, k, sizeof k);
            c->kept_packed_bytes = pb; c->kept_pq_bytes = qb; c->kept_n_reads = n_reads; c->kept_budget = saved - c->budget;
            if (d_bc) { (void)hipFree(d_bc); const uint64_t b = n_reads * 4 + 64; c->budget += b; c->kept_budget -= b; }
            return 0;
        }
        c->budget = saved;
    }
    (void)hipFree(d_packed); (void)hipFree(d_boff); (void)hipFree(d_len); (void)hipFree(d_pq); (void)hipFree(d_poff); (void)hipFree(d_bc);
    return rc;
}

int dfk_count(dfk_ctx* c, const uint8_t* packed, const uint64_t* base_off, const uint32_t* read_len,
              const uint8_t* pq, const uint64_t* pq_off, const int32_t* bc, uint64_t n_reads)
{
    return guarded([&]() -> int { return count_host(c, packed, base_off, read_len, pq, pq_off, bc, nullptr, 0, n_reads); });
}

int dfk_count_bci(dfk_ctx* c, const uint8_t* packed, const uint64_t* base_off, const uint32_t* read_len,
                  const uint8_t* pq, const uint64_t* pq_off, const int64_t* bci, uint64_t n_bci, uint64_t n_reads)
{
    return guarded([&]() -> int {
        if (!bci) return fail(DFK_E_ARG, "null barcode index");
        return count_host(c, packed, base_off, read_len, pq, pq_off, nullptr, bci, n_bci, n_reads);
    });
}

int dfk_good_lens(dfk_ctx* c, uint32_t* out, uint64_t cap)
{
    return guarded([&]() -> int {
    if (!c || !c->have) return fail(DFK_E_STATE, "no completed count");
    if (cap < c->n_reads) return fail(DFK_E_ARG, "buffer too small");
    if (c->n_reads && !c->good_len.p) return fail(DFK_E_STATE, "the trimmed lengths were given back when the reads were pathed (dfk_paths_build): fetch them before");
    HIP_TRY(hipSetDevice(c->device));
    if (c->n_reads) HIP_TRY(hipMemcpy(out, c->good_len.p, c->n_reads * 4, hipMemcpyDeviceToHost));
    return 0;
    });
}

int dfk_spectrum(dfk_ctx* c, const int64_t** hist, uint64_t* nbins)
{
    if (!c || !c->have) return fail(DFK_E_STATE, "no completed count");
    *hist = c->hist.data(); *nbins = c->hist.size();
    return 0;
}

int dfk_spectrum_json(dfk_ctx* c, char* out, uint64_t cap, uint64_t* need)
{
    return guarded([&]() -> int {
    if (!c || !c->have) return fail(DFK_E_STATE, "no completed count");
    // WriteHistToJson<int64_t> (10X/MakeHist.cc:67-92) as called from WriteKmerSpectrum
    std::string s = "{\n\t\"description\": \"kmer_count\",\n\t\"stage\": \"DF\",\n\t\"binsize\": 1,\n\t\"min\": 0,\n";
    s += "\t\"max\": " + std::to_string((long long)c->hist.size() - 1) + ",\n";
    s += "\t\"numbins\": " + std::to_string(c->hist.size()) + ",\n\t\"vals\": [";
    for (size_t i = 0; i < c->hist.size(); ++i) { s += std::to_string(c->hist[i]); if (i + 1 != c->hist.size()) s += ","; }
    s += "]\n}\n";
    if (need) *need = s.size();
    if (out && cap) { size_t n = std::min<size_t>(cap, s.size()); memcpy(out, s.data(), n); if (n < cap) out[n] = 0; }
    return 0;
    });
}

int dfk_solid_count(dfk_ctx* c, uint64_t* n)
{
    if (!c || !c->have) return fail(DFK_E_STATE, "no completed count");
    *n = c->n_solid;
    return 0;
}

int dfk_solid_fetch(dfk_ctx* c, dfk_entry32* out, uint64_t cap, int pre)
{
    return guarded([&]() -> int {
    if (!c || !c->have) return fail(DFK_E_STATE, "no completed count");
    if (cap < c->n_solid) return fail(DFK_E_ARG, "buffer too small: %llu < %llu", (unsigned long long)cap, (unsigned long long)c->n_solid);
    HIP_TRY(hipSetDevice(c->device));
    std::vector<dfk_entry32>* v = nullptr;
    int rc = fetch_sorted(c, pre != 0, &v); if (rc) return rc;
    if (!v->empty()) memcpy(out, v->data(), v->size() * 32);
    return 0;
    });
}

int dfk_solid_fetch_unsorted(dfk_ctx* c, dfk_entry32* out, uint64_t cap, int pre)
{
    return guarded([&]() -> int {
    if (!c || !c->have) return fail(DFK_E_STATE, "no completed count");
    if (cap < c->n_solid) return fail(DFK_E_ARG, "buffer too small: %llu < %llu", (unsigned long long)cap, (unsigned long long)c->n_solid);
    if (pre && !(c->cfg.flags & DFK_F_KEEP_PRE_ADJ)) return fail(DFK_E_STATE, "pre-adjacency view needs DFK_F_KEEP_PRE_ADJ");
    HIP_TRY(hipSetDevice(c->device));
    uint64_t at = 0;
    for (const dfk_ctx::Part& pt : c->parts) {
        const DevBuf& src = pre ? pt.pre : pt.buf;
        if (pt.n) HIP_TRY(hipMemcpy(out + at, src.p, pt.n * 32, hipMemcpyDeviceToHost));
        at += pt.n;
    }
    return 0;
    });
}

int dfk_solid_digest(dfk_ctx* c, int pre, uint64_t* digest)
{
    return guarded([&]() -> int {
    if (!c || !digest) return fail(DFK_E_ARG, "null argument");
    if (!c->have) return fail(DFK_E_STATE, "no completed count");
    if (pre && !(c->cfg.flags & DFK_F_KEEP_PRE_ADJ)) return fail(DFK_E_STATE, "pre-adjacency view needs DFK_F_KEEP_PRE_ADJ");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf d; int rc = c->alloc(d, 16, "digest"); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d.p, 0, 16, c->stream));
    for (const dfk_ctx::Part& pt : c->parts) {
        const DevBuf& src = pre ? pt.pre : pt.buf;
        if (!pt.n) continue;
        const unsigned grid = (unsigned)std::min<uint64_t>((pt.n + 255) / 256, 8192);
        hipLaunchKernelGGL(k_digest, dim3(grid), dim3(256), 0, c->stream, (const uint4*)src.p, (uint64_t)pt.n, (unsigned long long*)d.p);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(digest, d.p, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->release(d);
    return 0;
    });
}

int dfk_write_kvec(dfk_ctx* c, const char* path, int flags)
{
    return guarded([&]() -> int {
    if (!c || !c->have) return fail(DFK_E_STATE, "no completed count");
    if (!path) return fail(DFK_E_ARG, "null path");
    const int pre = flags & DFK_KVEC_PRE_ADJ;
    if (pre && !(c->cfg.flags & DFK_F_KEEP_PRE_ADJ)) return fail(DFK_E_STATE, "pre-adjacency view needs DFK_F_KEEP_PRE_ADJ");
    HIP_TRY(hipSetDevice(c->device));
    if (!(flags & DFK_KVEC_SORTED)) {
        // device order, as the reference's own kmers.kvec is in thread-arrival order (BuildReadQGraph48.cc:287-288,
        // ReadPather.h:406-418): no host copy of the dictionary, no sort
        const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd < 0) return fail(DFK_E_ARG, "cannot open %s", path);
        const uint64_t n = c->n_solid;
        char head[16]; memcpy(head, "BINWRITE", 8); memcpy(head + 8, &n, 8);
        int rc = pwrite(fd, head, 16, 0) == 16 ? 0 : fail(DFK_E_ARG, "short write to %s", path);
        if (!rc && ftruncate(fd, (off_t)(16 + 32 * n)) != 0) rc = fail(DFK_E_ARG, "cannot size %s", path);
        if (!rc) rc = write_parts_unsorted(c, fd, pre != 0);
        if (close(fd) != 0 && !rc) rc = fail(DFK_E_ARG, "short write to %s", path);
        return rc;
    }
    std::vector<dfk_entry32>* v = nullptr;
    int rc = fetch_sorted(c, pre != 0, &v); if (rc) return rc;
    FILE* f = fopen(path, "wb");
    if (!f) return fail(DFK_E_ARG, "cannot open %s", path);
    uint64_t n = v->size();
    bool ok = fwrite("BINWRITE", 1, 8, f) == 8 && fwrite(&n, 8, 1, f) == 1 && (n == 0 || fwrite(v->data(), 32, n, f) == n);
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : fail(DFK_E_ARG, "short write to %s", path);
    });
}

int dfk_write_kvec_part(dfk_ctx* c, const char* path, int flags, uint64_t first_entry, uint64_t total_entries)
{
    return guarded([&]() -> int {
    if (!c || !c->have) return fail(DFK_E_STATE, "no completed count");
    if (!path) return fail(DFK_E_ARG, "null path");
    if (flags & DFK_KVEC_SORTED) return fail(DFK_E_ARG, "the shares of several ranks are written in device order");
    const int pre = flags & DFK_KVEC_PRE_ADJ;
    if (pre && !(c->cfg.flags & DFK_F_KEEP_PRE_ADJ)) return fail(DFK_E_STATE, "pre-adjacency view needs DFK_F_KEEP_PRE_ADJ");
    if (first_entry + c->n_solid > total_entries) return fail(DFK_E_ARG, "this rank's %llu entries from %llu on do not fit %llu", (unsigned long long)c->n_solid,
                                                              (unsigned long long)first_entry, (unsigned long long)total_entries);
    HIP_TRY(hipSetDevice(c->device));
    const int fd = open(path, O_WRONLY | O_CREAT, 0666);
    if (fd < 0) return fail(DFK_E_ARG, "cannot open %s", path);
    int rc = 0;
    if (first_entry == 0) {
        char head[16]; memcpy(head, "BINWRITE", 8); memcpy(head + 8, &total_entries, 8);
        if (pwrite(fd, head, 16, 0) != 16) rc = fail(DFK_E_ARG, "short write to %s", path);
    }
    if (!rc && ftruncate(fd, (off_t)(16 + 32 * total_entries)) != 0) rc = fail(DFK_E_ARG, "cannot size %s", path);   // every rank: the same size
    if (!rc) rc = write_parts_unsorted(c, fd, pre != 0, 16 + 32 * first_entry);
    if (close(fd) != 0 && !rc) rc = fail(DFK_E_ARG, "short write to %s", path);
    return rc;
    });
}

int dfk_get_stats(dfk_ctx* c, dfk_stats* out)
{
    if (!c || !out) return fail(DFK_E_ARG, "null argument");
    if (c->d_resident && hipSetDevice(c->device) == hipSuccess) {           // reserved[4]: sweeps that went ahead of a k_count launch still being placed (k_gate timed out)
        unsigned int t = 0;
        if (hipMemcpy(&t, c->d_resident + 1, 4, hipMemcpyDeviceToHost) == hipSuccess) c->st.reserved[4] = t;
    }
    c->st.reserved[5] = c->held;                                             // device bytes the context holds right now
    *out = c->st;
    return 0;
}

} // extern "C"

#include "dfk_shard.inc"
#include "dfk_graph.inc"
#include "dfk_paths.inc"
#include "dfk_paths_shard.inc"
#include "dfk_pbf.inc"
