// superplus_amd/csrc/dfk_graph_kernels.h -- SURVEY 8(f)-1: the unipath edges of the solid k-mer graph, on the device
// dictionary (gfx950, wave64).  What the reference does in EdgeBuilder / buildEdges
// (paths/long/BuildReadQGraph48.cc:320-530) with one thread per hash-set stripe walking edges through a hopscotch
// dictionary is done here by one lane per edge END: pointer chasing through an HBM index, bound by the latency of
// dependent random reads, not by bandwidth.  Integer work only.
//
// Vocabulary: an entry's k-mer is stored canonical (F <= rc(F)); a walk holds a k-mer in the orientation it travels in
// and the entry's context byte turned to that orientation (KMerContext::rc = bit reversal).
#pragma once
#include "dfk_kernels.h"

namespace dfk {

// The dictionary is one dense array of 32-byte entries per counting pass; entries are addressed by one global index.
constexpr int GRAPH_MAX_PARTS = 96;
struct PartTable {
    uint32_t n_parts;
    uint32_t pad;
    uint64_t start[GRAPH_MAX_PARTS + 1];          // global index of each part's first entry; start[n_parts] = n_solid
    uint4* ptr[GRAPH_MAX_PARTS];
};

// The table as the kernels use it: in LDS, with the part of every 256th of the index range looked up first -- an entry's
// address is then three or four LDS reads.  (Searched where the kernel arguments lie, with a lane's own index, it was seven
// dependent vector loads: half the time of a walk's step.)
struct PartLds {
    uint64_t start[GRAPH_MAX_PARTS + 1];
    uint4* ptr[GRAPH_MAX_PARTS];
    uint8_t first[256];                            // the part that holds entry (slot << shift)
    uint32_t shift, n_parts;
};
__device__ __forceinline__ void part_lds_init(const PartTable& a, PartLds& L)      // by every thread of the block, before anything else
{
    const uint32_t np = a.n_parts;
    for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) { L.start[i] = a.start[i]; L.ptr[i] = a.ptr[i]; }
    const uint64_t n = a.start[np];
    const uint32_t sh = n >> 8 ? 64u - (uint32_t)__clzll((long long)n) - 8u : 0u;  // n >> sh < 256
    if (threadIdx.x == 0) { L.start[np] = n; L.shift = sh; L.n_parts = np; }
    __syncthreads();
    for (uint32_t slot = threadIdx.x; slot < 256; slot += blockDim.x) {
        const uint64_t g = (uint64_t)slot << sh;
        uint32_t lo = 0, hi = np;                     // start[lo] <= g < start[hi], or the last part
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (L.start[mid] <= g) lo = mid; else hi = mid; }
        L.first[slot] = (uint8_t)lo;
    }
    __syncthreads();
}
__device__ __forceinline__ uint4* entry_ptr(const PartLds& L, uint64_t g)
{
    uint32_t p = L.first[(g >> L.shift) & 255u];
    while (p + 1 < L.n_parts && L.start[p + 1] <= g) ++p;
    return L.ptr[p] + 2 * (g - L.start[p]);
}

// k-mer algebra on 2K-bit big-endian values (base 0 most significant: KMer<K>'s own order, kmers/KMer.h:154-160)
template <int K> __device__ __forceinline__ u128 kmer_of_entry(const uint4 a)
{
    const u128 kw{(uint64_t)a.z | ((uint64_t)a.w << 32), (uint64_t)a.x | ((uint64_t)a.y << 32)};   // lo = w1, hi = w0
    return shr128(kw, 128 - KTraits<K>::BITS);
}
template <int K> __device__ __forceinline__ u128 kmer_rc(u128 F)
{
    const u128 m = KTraits<K>::mask();
    const u128 nf{~F.lo & m.lo, ~F.hi & m.hi};
    const u128 top = shl128(nf, 128 - KTraits<K>::BITS);
    return u128{rev2_64(top.hi), rev2_64(top.lo)};
}
template <int K> __device__ __forceinline__ u128 kmer_succ(u128 F, uint32_t b)            // KMer::toSuccessor
{ const u128 m = KTraits<K>::mask(); u128 v = shl128(F, 2); v.lo = (v.lo & m.lo) | b; v.hi &= m.hi; return v; }
template <int K> __device__ __forceinline__ u128 kmer_pred(u128 F, uint32_t b)            // KMer::toPredecessor
{
    u128 v = shr128(F, 2);
    constexpr int TOP = KTraits<K>::BITS - 2;
    if (TOP >= 64) v.hi |= (uint64_t)b << (TOP - 64); else v.lo |= (uint64_t)b << TOP;
    return v;
}
__device__ __forceinline__ bool eq128(u128 a, u128 b) { return a.lo == b.lo && a.hi == b.hi; }
template <int K> __device__ __forceinline__ uint32_t kmer_base(u128 F, int i)             // base i, 0 = first
{ const int sh = 2 * (K - 1 - i); return (uint32_t)(sh >= 64 ? (F.hi >> (sh - 64)) : (F.lo >> sh)) & 3u; }

// context byte: low nibble successors, high nibble predecessors, bit = 1 << base (kmers/KMerContext.h:36-78)
__device__ __forceinline__ uint32_t n_succ(uint32_t c) { return __popc(c & 15u); }
__device__ __forceinline__ uint32_t n_pred(uint32_t c) { return __popc(c >> 4); }
__device__ __forceinline__ uint32_t one_succ(uint32_t c) { return (uint32_t)__ffs(c & 15u) - 1u; }
__device__ __forceinline__ uint32_t one_pred(uint32_t c) { return (uint32_t)__ffs(c >> 4) - 1u; }

// ---- the index: canonical k-mer -> global entry index.  Open addressing over u32 slots, linear probing, load <= 0.5;
// a slot names an entry and the key is read from the entry itself (the dictionary is the key store).
constexpr uint32_t GRAPH_EMPTY = 0xFFFFFFFFu;

// where a key's probe sequence starts and how it goes on.  The table has as many slots as its load asks for, not the next
// power of two (which at 3.1e9 k-mers would be 34 GB where 25 do).
__device__ __forceinline__ uint64_t index_home(uint64_t hash, uint64_t n_slots) { return __umul64hi(hash, n_slots); }
__device__ __forceinline__ uint64_t index_next(uint64_t s, uint64_t n_slots) { return s + 1 == n_slots ? 0 : s + 1; }
template <int K>
__global__ void __launch_bounds__(256)
k_graph_index(PartTable pt_arg, uint64_t n, uint32_t* __restrict__ index, uint64_t n_slots)
{
    __shared__ PartLds pt;
    part_lds_init(pt_arg, pt);
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (uint64_t)gridDim.x * 256) {
        const uint4 a = *entry_ptr(pt, g);
        const uint64_t w0 = (uint64_t)a.x | ((uint64_t)a.y << 32), w1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
        uint64_t s = index_home(set_hash(w0, w1), n_slots);
        while (atomicCAS(&index[s], GRAPH_EMPTY, (uint32_t)g) != GRAPH_EMPTY) s = index_next(s, n_slots);
    }
}

// The entry of k-mer `v` (any orientation) and its context as seen travelling in v's orientation: EdgeBuilder::lookup
// (BuildReadQGraph48.cc:467-478).  Returns GRAPH_EMPTY if the k-mer is not in the dictionary (cannot happen for a
// neighbour named by a context bit after recomputeAdjacencies; callers treat it as "stop").
template <int K>
__device__ __forceinline__ uint32_t graph_lookup(const PartLds& pt, const uint32_t* __restrict__ index, uint64_t n_slots, u128 v,
                                                 uint32_t* ctx, bool* is_pal, bool* is_rev = nullptr)
{
    const u128 R = kmer_rc<K>(v);
    const bool rev = lt128(R, v);
    *is_pal = eq128(R, v);
    if (is_rev) *is_rev = rev;
    const u128 c = rev ? R : v;
    const u128 kw = shl128(c, 128 - KTraits<K>::BITS);
    const uint64_t w0 = kw.hi, w1 = kw.lo;
    uint64_t s = index_home(set_hash(w0, w1), n_slots);
    for (uint32_t guard = 0; guard < 1u << 20; ++guard) {
        const uint32_t g = index[s];
        if (g == GRAPH_EMPTY) return GRAPH_EMPTY;
        const uint4* e = entry_ptr(pt, g);
        const uint4 a = e[0];
        if (((uint64_t)a.x | ((uint64_t)a.y << 32)) == w0 && ((uint64_t)a.z | ((uint64_t)a.w << 32)) == w1) {
            const uint32_t cx = e[1].y >> 24;
            *ctx = rev ? ctx_rc(cx) : cx;
            return g;
        }
        s = index_next(s, n_slots);
    }
    return GRAPH_EMPTY;
}

// ---- classification (buildEdge, :326-336 with upstream/downstreamExtensionPossible :399-419)
// While the edges are being found, an entry's second 16 bytes hold what a walk needs to step to its neighbours WITHOUT
// going through the index again (the classification looks the neighbours up anyway), as two 8-byte halves:
//   (x, y): x = entry of the k-mer that follows this one in its canonical orientation (its single successor), if looked up
//           y = context byte << 24 | 0xFFFFFF (the count is not needed any more: the offset will take its place)
//   (z, w): z = entry of the k-mer that precedes it (its single predecessor), if looked up
//           w = kind (2 bits) | GL_DOWN | GL_UP (x / z valid) | GL_DOWN_REV | GL_UP_REV (that neighbour's canonical form is
//               the reverse complement of the k-mer as this one's orientation reads it) | GL_PAL | GL_PLACED | 0xFFFFFF << 8
// A walk step is then ONE dependent 16-byte read (the next entry's context and links) instead of an index probe and the
// entry's 32 bytes behind it.  A walker that passes an entry FORWARD (down its canonical orientation) has no more use for
// the down link and leaves its STAMP in that half: (x, y) = (its number in the list of ends, context << 24 | its step); one
// that passes it REVERSED stamps (z, w) = (its number, flags | step << 8).  k_graph_place turns stamps into (edge, offset).
enum : uint32_t { GK_INTERIOR = 0, GK_END_DOWN = 1,      // first k-mer of an edge read in its canonical orientation
                  GK_END_UP = 2,                         // last k-mer of an edge read in its canonical orientation: walked as its reverse complement
                  GK_SINGLE = 3,                         // an edge of one k-mer: a palindrome, or no extension either way
                  GL_DOWN = 4, GL_UP = 8, GL_DOWN_REV = 16, GL_UP_REV = 32, GL_PAL = 64, GL_PLACED = 128,
                  STAMP_NONE = 0xFFFFFFu };

template <int K>
__global__ void __launch_bounds__(256)
k_graph_classify(PartTable pt_arg, uint64_t n, const uint32_t* __restrict__ index, uint64_t n_slots, unsigned long long* __restrict__ n_ends)
{
    __shared__ PartLds pt;
    part_lds_init(pt_arg, pt);
    unsigned long long mine = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (uint64_t)gridDim.x * 256) {
        uint4* e = entry_ptr(pt, g);
        const u128 F = kmer_of_entry<K>(e[0]);
        uint4 b = e[1];
        const uint32_t ctx = b.y >> 24;
        uint32_t kind, links = 0, down_to = 0xFFFFFFFFu, up_to = 0xFFFFFFFFu;
        if (eq128(kmer_rc<K>(F), F)) { kind = GK_SINGLE; links = GL_PAL; }
        else {
            bool up = false, down = false, pal, rev;
            uint32_t c2;
            if (n_pred(ctx) == 1) {
                const u128 p = kmer_pred<K>(F, one_pred(ctx));
                const uint32_t g2 = graph_lookup<K>(pt, index, n_slots, p, &c2, &pal, &rev);
                up = !pal && g2 != GRAPH_EMPTY && n_succ(c2) == 1;
                if (g2 != GRAPH_EMPTY) { up_to = g2; links |= GL_UP | (rev ? GL_UP_REV : 0u); }
            }
            if (n_succ(ctx) == 1) {
                const u128 s = kmer_succ<K>(F, one_succ(ctx));
                const uint32_t g2 = graph_lookup<K>(pt, index, n_slots, s, &c2, &pal, &rev);
                down = !pal && g2 != GRAPH_EMPTY && n_pred(c2) == 1;
                if (g2 != GRAPH_EMPTY) { down_to = g2; links |= GL_DOWN | (rev ? GL_DOWN_REV : 0u); }
            }
            kind = up ? (down ? GK_INTERIOR : GK_END_UP) : (down ? GK_END_DOWN : GK_SINGLE);
        }
        b.x = down_to; b.y |= STAMP_NONE; b.z = up_to; b.w = kind | links | (STAMP_NONE << 8);
        e[1] = b;
        mine += kind != GK_INTERIOR;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_ends, mine);
}

// A walker's position: the entry it stands on, whether it reads that entry's k-mer reverse-complemented, and the entry's
// context and links (its second 16 bytes).
struct WalkPos { uint32_t g; bool rc; uint4 b; };
__device__ __forceinline__ uint32_t walk_ctx(const WalkPos& p) { const uint32_t c = p.b.y >> 24; return p.rc ? ctx_rc(c) : c; }
// An entry's second half as the other walkers' stamps have left it.  The stamps are WRITTEN at device scope (the 8 XCDs' L2s
// do not see each other's writes otherwise); the read is an ordinary one: a copy some microseconds old out of this XCD's
// L2 can only lack a stamp -- a meeting missed, the walk goes on to the next stamped entry or to the far end -- and a read
// at device scope would give up the L2 hits on the neighbouring entries (consecutive k-mers share their minimizer, so their
// entries share a bucket: measured, twice the time per step).
__device__ __forceinline__ uint4 links_now(const uint4* e) { return e[1]; }
// the stamp of the walker that would come the OTHER way through p, if it has been here: its number and its step
__device__ __forceinline__ bool met_other(const WalkPos& p, uint32_t* t, uint32_t* step)
{
    const uint32_t st = p.rc ? (p.b.y & STAMP_NONE) : (p.b.w >> 8);
    *t = p.rc ? p.b.x : p.b.z; *step = st;
    return st != STAMP_NONE;
}
__device__ __forceinline__ void stamp(const PartLds& pt, const WalkPos& p, uint32_t t, uint32_t step)
{
    unsigned long long* q = reinterpret_cast<unsigned long long*>(entry_ptr(pt, p.g) + 1);
    const unsigned long long v = p.rc ? ((unsigned long long)((p.b.w & 0xFFu) | (step << 8)) << 32 | t)
                                      : ((unsigned long long)((p.b.y & 0xFF000000u) | step) << 32 | t);
    __hip_atomic_store(q + (p.rc ? 1 : 0), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// To the single successor in the walker's orientation, over the stored link; false = no such k-mer in the dictionary.
__device__ __forceinline__ bool walk_follow(const PartLds& pt, const WalkPos& p, WalkPos* q)
{
    const uint32_t w = p.b.w;
    if (n_succ(walk_ctx(p)) != 1 || !(w & (p.rc ? GL_UP : GL_DOWN))) return false;
    q->g = p.rc ? p.b.z : p.b.x;
    // travelling reversed, the next k-mer is the canonical predecessor read backwards: it reads forwards iff ITS canonical form
    // is the reverse complement of the predecessor as written
    q->rc = p.rc ? !(w & GL_UP_REV) : (w & GL_DOWN_REV) != 0u;
    q->b = links_now(entry_ptr(pt, q->g));
    return true;
}
// dense list of the entries that are edge ends (kind != interior); `want_null_interior`: the list of interior entries
// still without an edge instead (the members of branch-free cycles)
__global__ void __launch_bounds__(256)
k_graph_list(PartTable pt_arg, uint64_t n, bool want_null_interior, uint32_t* __restrict__ list, uint64_t cap, unsigned long long* __restrict__ n_list)
{
    __shared__ PartLds pt;
    part_lds_init(pt_arg, pt);
    __shared__ uint32_t found[256 * 8];
    __shared__ uint32_t n_found;
    __shared__ unsigned long long at;
    const int lane = threadIdx.x & 63;
    for (uint64_t g0 = (uint64_t)blockIdx.x * 2048; g0 < n; g0 += (uint64_t)gridDim.x * 2048) {
        if (threadIdx.x == 0) n_found = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t g = g0 + 256ull * j + threadIdx.x;
            bool hit = false;
            if (g < n) { const uint32_t w = entry_ptr(pt, g)[1].w; hit = want_null_interior ? ((w & (3u | GL_DOWN | GL_UP | GL_PLACED)) == (GK_INTERIOR | GL_DOWN | GL_UP)) : ((w & 3u) != GK_INTERIOR); }
            const unsigned long long mk = __ballot(hit);
            uint32_t w = 0;
            if (lane == 0 && mk) w = atomicAdd(&n_found, (uint32_t)__popcll(mk));
            w = __builtin_amdgcn_readfirstlane(w);
            if (hit) found[w + __popcll(mk & ((1ull << lane) - 1ull))] = (uint32_t)g;
        }
        __syncthreads();
        const uint32_t m = n_found;
        if (threadIdx.x == 0) at = m ? atomicAdd(n_list, (unsigned long long)m) : 0ull;
        __syncthreads();
        const unsigned long long base = at;
        for (uint32_t t = threadIdx.x; t < m; t += 256) if (base + t < cap) list[base + t] = found[t];
        __syncthreads();
    }
}

// the pad word back to 0 (KmerDictEntry's padding), the tempBC word back to -1 where a cycle's walk did not pass
__global__ void __launch_bounds__(256)
k_graph_finalize(PartTable pt_arg, uint64_t n)
{
    __shared__ PartLds pt;
    part_lds_init(pt_arg, pt);
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (uint64_t)gridDim.x * 256) {
        uint4* e = entry_ptr(pt, g);
        uint4 b = e[1];
        if (b.w != 0u || b.z != 0xFFFFFFFFu) { b.w = 0u; b.z = 0xFFFFFFFFu; e[1] = b; }
    }
}

// ---- edges
// One record per canonical edge, in the order the owners reserved them (arbitrary: buildHBVFromEdges sorts).
struct EdgeRec {
    uint32_t g_start;      // entry the owner's walk starts from
    uint32_t n;            // k-mers on the edge
    uint64_t byte_off;     // of its 2-bit bases in the edge store (byte aligned, LSB-first like a .fastb)
    uint32_t flags;        // ER_*; after k_graph_walk_write bit 31 = the edge is stored reverse-complemented relative to the walk
    uint32_t g_last;       // entry the walk ends on (what decides the orientation of an even-length edge, with ER_LAST_RC)
};
constexpr uint32_t ER_START_RC = 1u, ER_CYCLE = 2u, ER_LAST_RC = 4u, ER_STORED_REV = 0x80000000u;

// wave-aggregated reservation: every lane with `mine` gets an edge number and room for `bytes` bytes
__device__ __forceinline__ void reserve_edge(bool mine, uint64_t bytes, unsigned long long* __restrict__ ctr /* [0] edges, [1] bytes */,
                                             uint64_t* edge_no, uint64_t* byte_off)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long mk = __ballot(mine);
    if (!mk) return;
    // exclusive prefix of the byte counts over the wave
    uint64_t incl = mine ? bytes : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    const uint64_t total = __shfl(incl, 63, 64);
    unsigned long long e0 = 0, b0 = 0;
    if (lane == 0) { e0 = atomicAdd(&ctr[0], (unsigned long long)__popcll(mk)); b0 = atomicAdd(&ctr[1], (unsigned long long)total); }
    e0 = uniform64((uint64_t)e0);
    b0 = uniform64((uint64_t)b0);
    *edge_no = e0 + __popcll(mk & ((1ull << lane) - 1ull));
    *byte_off = b0 + incl - (mine ? bytes : 0);
}

// Pass A: the two ends of an edge walk towards each other (EdgeBuilder::extend, :436-456), stamping what they pass, until
// each meets the other's stamp or -- if the other has not started, or its stamps were not seen in time -- the far end.
// Either way a walker then knows the edge's length and its other end: meeting a stamp (t', s') after a entries of its own,
// n = a + s' + 1 and the other end is ends[t'].  The end whose canonical k-mer is the smaller of the two owns the edge and
// reserves its record; a one-k-mer edge owns itself.  (The reference builds an edge from whichever end its thread meets
// first and throws away the walk that comes out in REV form; owning by k-mer order makes exactly one lane record each edge.)
// info[t] = (edge, 1) for the owner, and (edge, 0) for the other end when the owner met it: every stamp an entry may be left
// with alone leads to its edge (an entry only the non-owner reached is one the owner stopped short of: they met).
// Edge lengths are spread exponentially: lanes that have finished take the next ends off a cursor as soon as WALK_REFILL
// of them are idle; the wave makes its reservations at the same point, where all its lanes meet.
constexpr int WALK_REFILL = 16;
constexpr uint32_t NO_EDGE = 0xFFFFFFFFu;
template <int K>
__global__ void __launch_bounds__(256)
k_graph_walk_count(PartTable pt_arg, const uint32_t* __restrict__ ends, uint64_t n_ends,
                   EdgeRec* __restrict__ recs, uint64_t rec_cap, uint2* __restrict__ info, unsigned long long* __restrict__ ctr /* [4]: the cursor */,
                   uint32_t max_steps, unsigned int* __restrict__ bad)
{
    __shared__ PartLds pt;
    part_lds_init(pt_arg, pt);
    const int lane = threadIdx.x & 63;
    bool active = false, done = false, owner = false, dry = false;
    WalkPos p{0, false, uint4{0, 0, 0, 0}};
    uint32_t n = 1, g = 0, flags = 0, t = 0, t_other = NO_EDGE, g_other = 0;
    // the walk is over: the edge has n k-mers and its far end is entry `far` (read reversed by this walk: far_rc)
    auto finish = [&](uint32_t far, bool far_rc) {
        active = false; done = true; g_other = far;
        if (n > 0x00FFFFFFu) { atomicOr(bad, 2u); owner = false; return; }
        const uint4* e0 = entry_ptr(pt, g);
        const u128 F0 = kmer_of_entry<K>(e0[0]), Ff = kmer_of_entry<K>(entry_ptr(pt, far)[0]);
        owner = lt128(F0, Ff);
        if (owner && !((n + K - 1) & 1)) {                               // even length: the first and the last k-mer decide the stored form
            const u128 first = (flags & ER_START_RC) ? kmer_rc<K>(F0) : F0, last = far_rc ? kmer_rc<K>(Ff) : Ff;
            if (lt128(kmer_rc<K>(last), first)) flags |= ER_STORED_REV;  // rc(S) begins with rc(last k-mer)
        }
    };
    // arrived at p: the other walker's entry already? else it is ours -- and stamped on the way out, BEHIND the read of the
    // next entry: the step then waits for that read alone (the counter a wave waits on retires in order; a write-through
    // stamp ahead of the read doubled the time per step, and the longest edge's steps are what the kernel lasts)
    bool unstamped = false;
    auto arrive = [&]() {
        uint32_t t2, s2;
        if (met_other(p, &t2, &s2)) {
            atomicAdd(&ctr[6], (unsigned long long)(n - 1));             // (entries walked, for the trace)
            n += s2;                                                     // n counted ours + 1 on the way here
            t_other = t2;
            const uint32_t far = ends[t2];
            // the other walker reads its start forwards iff that is an END_DOWN; we come the other way
            finish(far, (entry_ptr(pt, far)[1].w & 3u) == GK_END_DOWN);
            return;
        }
        unstamped = true;
    };
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (idle == ~0ull || (!dry && __popcll(idle) >= WALK_REFILL)) {
            uint64_t eno = 0, off = 0;
            reserve_edge(done && owner, ((uint64_t)n + K - 1 + 3) / 4, ctr, &eno, &off);
            if (done && owner && eno < rec_cap) {
                recs[eno] = EdgeRec{g, n, off, flags, g_other};
                if (n == 1) entry_ptr(pt, g)[1].x = (uint32_t)eno;      // a one-k-mer edge: nobody walks it, the entry itself says which
                else { info[t] = uint2{(uint32_t)eno, 1u}; if (t_other != NO_EDGE) info[t_other] = uint2{(uint32_t)eno, 0u}; }
            }
            done = false;
            if (dry) break;                                                  // (only reached with every lane idle)
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&ctr[4], (unsigned long long)__popcll(idle));
            base = uniform64((uint64_t)base);
            dry = base + __popcll(idle) >= n_ends;
            const uint64_t tt = base + __popcll(idle & ((1ull << lane) - 1ull));
            if (!active && tt < n_ends) {
                t = (uint32_t)tt; g = ends[tt]; g_other = g;
                p = WalkPos{g, false, links_now(entry_ptr(pt, g))};
                n = 1; flags = 0; t_other = NO_EDGE;
                const uint32_t kind = p.b.w & 3u;
                if (kind == GK_SINGLE) { done = true; owner = true; }
                else { active = true; if (kind == GK_END_UP) { p.rc = true; flags = ER_START_RC; } arrive(); }
            }
            continue;
        }
        if (active) {
            if (n >= max_steps) { atomicOr(bad, 2u); active = false; }
            else {
                WalkPos q;
                const bool on = walk_follow(pt, p, &q);                   // (the read)
                if (unstamped) { stamp(pt, p, t, n - 1); unstamped = false; }
                // EdgeBuilder::extend's loop body (:436-456): the edge ends at p if it has several successors, an unknown or
                // palindromic one, or one with several predecessors
                if (on && !(q.b.w & GL_PAL) && n_pred(walk_ctx(q)) == 1) { p = q; ++n; arrive(); }
                else { atomicAdd(&ctr[6], (unsigned long long)n); finish(p.g, p.rc); }   // the far end, reached on foot
            }
        }
    }
}

// Pass B, over the entries: stamps -> (edge, step on the owner's walk, orientation on it), and the orientation of the stored
// edge where the middle base decides it.  The edge is stored in canonical form (addEdge :480-486): FWD or palindrome as
// walked, REV reverse-complemented, with the offsets counted from the other end.
//   getCanonicalForm (dna/CanonicalForm.h:32-46): odd length -> REV iff the middle base is G or T; even length ->
//   outside-in against the complement of the mirror base, which the first and the last k-mer of the walk decide
//   (they differ as k-mers, so one of their K positions differs) -- the owner's finish has done that.
// Between the two kernels an entry on an edge holds x = edge, y = context << 24 | step, z = -1, w = GL_PLACED | reversed on
// the owner's walk.
template <int K>
__global__ void __launch_bounds__(256)
k_graph_place(PartTable pt_arg, uint64_t n_entries, const uint2* __restrict__ info, EdgeRec* __restrict__ recs, unsigned int* __restrict__ bad)
{
    __shared__ PartLds pt;
    part_lds_init(pt_arg, pt);
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n_entries; g += (uint64_t)gridDim.x * 256) {
        uint4* e = entry_ptr(pt, g);
        uint4 b = e[1];
        if ((b.w & GL_PAL) || (b.w & 3u) == GK_SINGLE) continue;         // its own edge: k_graph_bases knows what to do
        const uint32_t sf = b.y & STAMP_NONE, sr = b.w >> 8;
        if (sf == STAMP_NONE && sr == STAMP_NONE) continue;              // no walker came by: on a branch-free cycle
        uint2 in{NO_EDGE, 0u};
        bool fwd = true;
        if (sf != STAMP_NONE) in = info[b.x];
        if (in.x == NO_EDGE && sr != STAMP_NONE) { in = info[b.z]; fwd = false; }
        if (in.x == NO_EDGE) { atomicOr(bad, 4u); continue; }
        const uint32_t n = recs[in.x].n, L = n + K - 1;
        const uint32_t s_mine = fwd ? sf : sr, s = in.y ? s_mine : n - 1 - s_mine;
        const bool rc = in.y ? !fwd : fwd;                               // the owner comes through the other way than a non-owner
        if (L & 1) {
            const uint32_t mid = L / 2, j = mid > (uint32_t)(K - 1) ? mid - (K - 1) : 0u;      // k-mer j of the walk holds base mid
            if (s == j) {
                const u128 F = kmer_of_entry<K>(e[0]);
                if (kmer_base<K>(rc ? kmer_rc<K>(F) : F, (int)(mid - j)) & 2u) atomicOr(&recs[in.x].flags, ER_STORED_REV);
            }
        }
        b.x = in.x; b.y = (b.y & 0xFF000000u) | s; b.z = 0xFFFFFFFFu; b.w = GL_PLACED | (rc ? 1u : 0u);
        e[1] = b;
    }
}
// ... then every entry puts its base(s) where the stored edge has them (2 bits into a zeroed store, 16 bases to the word) and
// takes its final form: x = edge, y = context << 24 | offset, z = -1, w = 0 -- the KDef of kmers/ReadPather.h:60-133.
template <int K>
__global__ void __launch_bounds__(256)
k_graph_bases(PartTable pt_arg, uint64_t n_entries, const EdgeRec* __restrict__ recs, unsigned int* __restrict__ store)
{
    __shared__ PartLds pt;
    part_lds_init(pt_arg, pt);
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n_entries; g += (uint64_t)gridDim.x * 256) {
        uint4* e = entry_ptr(pt, g);
        uint4 b = e[1];
        uint32_t s = 0; bool rc = false;
        if (b.w & GL_PLACED) { s = b.y & STAMP_NONE; rc = (b.w & 1u) != 0u; }
        else if (!((b.w & GL_PAL) || (b.w & 3u) == GK_SINGLE)) continue;
        const EdgeRec R = recs[b.x];
        const bool rev = (R.flags & ER_STORED_REV) != 0u;
        const uint32_t L = R.n + K - 1;
        const uint64_t bit0 = R.byte_off * 4;                            // in bases
        auto put = [&](uint32_t pos, uint32_t base) {                    // position on the owner's walk
            const uint64_t q = bit0 + (rev ? L - 1 - pos : pos);
            atomicOr(&store[q >> 4], (rev ? 3u - base : base) << (2 * (q & 15)));
        };
        const u128 F = kmer_of_entry<K>(e[0]), W = rc ? kmer_rc<K>(F) : F;
        if (s == 0) { for (int i = 0; i < K; ++i) put((uint32_t)i, kmer_base<K>(W, i)); }
        else put(K - 1 + s, (uint32_t)W.lo & 3u);
        b.y = (b.y & 0xFF000000u) | (rev ? R.n - 1 - s : s); b.z = 0xFFFFFFFFu; b.w = 0u;
        e[1] = b;
    }
}

// Pass C: what is still without an edge lies on a cycle without branches (simpleCircle, :338-365).  Every member
// walks the cycle until it meets a k-mer smaller than itself (then it is not the one) or comes back to itself: the
// smallest canonical k-mer of the cycle owns it, and the edge starts there in that k-mer's canonical orientation
// (canonicalizeCircle, :367-392).
template <int K>
__global__ void __launch_bounds__(256)
k_graph_cycles(PartTable pt_arg, const uint32_t* __restrict__ index, uint64_t n_slots, const uint32_t* __restrict__ members, uint64_t n_members,
               EdgeRec* __restrict__ recs, uint64_t rec_cap, unsigned long long* __restrict__ ctr, uint32_t max_steps,
               unsigned int* __restrict__ bad)
{
    __shared__ PartLds pt;
    part_lds_init(pt_arg, pt);
    const uint64_t rounds = (n_members + (uint64_t)gridDim.x * 256 - 1) / ((uint64_t)gridDim.x * 256);
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t t = (r * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        bool owner = false; uint32_t n = 1, g = 0;
        if (t < n_members) {
            g = members[t];
            const uint4* e = entry_ptr(pt, g);
            const u128 F = kmer_of_entry<K>(e[0]);
            uint32_t ctx = e[1].y >> 24;
            u128 cur = F;
            owner = true;
            for (;;) {
                if (n >= max_steps) { atomicOr(bad, 2u); owner = false; break; }
                if (n_succ(ctx) != 1) { atomicOr(bad, 1u); owner = false; break; }       // (a member of a branch-free cycle has one successor)
                const u128 nxt = kmer_succ<K>(cur, one_succ(ctx));
                uint32_t c2; bool pal;
                const uint32_t g2 = graph_lookup<K>(pt, index, n_slots, nxt, &c2, &pal);
                if (g2 == GRAPH_EMPTY) { owner = false; break; }
                if (g2 == g) break;                                          // back at the start: the whole cycle seen
                if (lt128(kmer_of_entry<K>(entry_ptr(pt, g2)[0]), F)) { owner = false; break; }
                cur = nxt; ctx = c2; ++n;
            }
        }
        uint64_t eno = 0, off = 0;
        reserve_edge(owner, ((uint64_t)n + K - 1 + 3) / 4, ctr, &eno, &off);
        if (owner && eno < rec_cap) recs[eno] = EdgeRec{g, n, off, ER_CYCLE, g};
    }
}

// (Cycles only.)  The owner walks its edge again and writes it: the bases into the edge store, and into every k-mer's entry
// the edge number and the k-mer's offset on it (KDef::set, :488-491 -- as in the reference the offset takes the
// place of the count, kmers/ReadPather.h:122-127).  The edge is stored in canonical form (addEdge :480-486): FWD or
// palindrome as walked, REV reverse-complemented, with the offsets counted from the other end.
//   getCanonicalForm (dna/CanonicalForm.h:32-46): odd length -> REV iff the middle base is G or T; even length ->
//   outside-in against the complement of the mirror base, which the first and the last k-mer of the walk decide
//   (they differ as k-mers, so one of their K positions differs).
// The store must be zero where no base is written (the last byte of an edge); the host clears it.  Lanes take edges off a
// cursor like the counting walk's.
template <int K>
__global__ void __launch_bounds__(256)
k_graph_walk_write(PartTable pt_arg, EdgeRec* __restrict__ recs, uint64_t e_lo, uint64_t e_hi,
                   uint8_t* __restrict__ store, unsigned long long* __restrict__ cursor, unsigned int* __restrict__ bad)
{
    __shared__ PartLds pt;
    part_lds_init(pt_arg, pt);
    // The walk is n - 1 steps along the links, whatever lies beyond the ends (the entry past an end may have been marked
    // by its own edge's writer already).  No entry is met twice on the way: with K even neither a k-mer followed by its
    // own reverse complement nor an edge equal to its reverse complement exists (the middle would be a palindrome, and
    // palindromes are edges of their own), so marking an entry as it is left never hides a link this walk still needs.
    static_assert(K % 2 == 0, "the walks rely on K being even");
    const int lane = threadIdx.x & 63;
    const uint64_t n_todo = e_hi - e_lo;
    bool active = false, dry = false, rev = false;
    WalkPos p{0, false, uint4{0, 0, 0, 0}};
    uint64_t eno = 0;
    uint8_t* out = nullptr;
    uint32_t n = 0, L = 0, s = 0, pre_left = 0, pos = 0, acc = 0, g_start = 0, rflags = 0;

    // position `pos` of the walk's sequence goes to q = rev ? L-1-pos : pos, as the base or its complement; bytes fill up in
    // one direction and are stored whole (a lane owns whole bytes: edges are byte aligned)
    auto emit = [&](uint32_t base) {
        const uint32_t q = rev ? L - 1 - pos : pos, v = rev ? 3u - base : base;
        acc |= v << (2 * (q & 3));
        if ((q & 3) == (rev ? 0u : 3u)) { out[q >> 2] = (uint8_t)acc; acc = 0; }
        ++pos;
    };
    auto mark = [&](const WalkPos& at, uint32_t st) {                    // entry <- (edge, offset); its links have been used
        uint4 b = at.b;
        const uint32_t off = rev ? n - 1 - st : st;
        b.x = (uint32_t)eno;
        b.y = (b.y & 0xFF000000u) | (off & 0xFFFFFFu);
        b.z = 0xFFFFFFFFu;
        b.w = GL_PLACED;
        entry_ptr(pt, at.g)[1] = b;
    };
    auto begin_write = [&]() {                                           // orientation known: back to the start, the first K bases
        const uint4* e0 = entry_ptr(pt, g_start);
        const u128 F0 = kmer_of_entry<K>(e0[0]);
        p = WalkPos{g_start, (rflags & ER_START_RC) != 0u, e0[1]};
        const u128 first = p.rc ? kmer_rc<K>(F0) : F0;
        pos = 0; acc = 0; s = 1; pre_left = 0;
        for (int i = 0; i < K; ++i) emit(kmer_base<K>(first, i));
    };

    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (idle == ~0ull || (!dry && __popcll(idle) >= WALK_REFILL)) {
            if (dry) break;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(cursor, (unsigned long long)__popcll(idle));
            base = uniform64((uint64_t)base);
            dry = base + __popcll(idle) >= n_todo;
            const uint64_t t = base + __popcll(idle & ((1ull << lane) - 1ull));
            if (!active && t < n_todo) {
                eno = e_lo + t;
                const EdgeRec R = recs[eno];
                n = R.n; L = n + K - 1; g_start = R.g_start; rflags = R.flags;
                out = store + R.byte_off;
                active = true; rev = false;
                const uint4* e0 = entry_ptr(pt, g_start);
                p = WalkPos{g_start, (rflags & ER_START_RC) != 0u, e0[1]};
                pre_left = 0;
                // ---- orientation of the stored edge.  Odd length: REV iff the middle base is G or T; even length: outside-in
                // against the complement of the mirror base, which the first and the last k-mer decide (they differ as k-mers).
                if (n > 1) {
                    const u128 F0 = kmer_of_entry<K>(e0[0]);
                    const u128 first = p.rc ? kmer_rc<K>(F0) : F0;
                    // the last k-mer in walk orientation: known from the counting walk, or (cycles, rare) found by walking
                    WalkPos lastp = p;
                    if (rflags & ER_CYCLE) {
                        for (uint32_t k = 1; k < n; ++k) { WalkPos q; if (!walk_follow(pt, lastp, &q)) { atomicOr(bad, 1u); break; } lastp = q; }
                    } else { lastp.g = R.g_last; lastp.rc = (rflags & ER_LAST_RC) != 0u; }
                    const u128 Fl = kmer_of_entry<K>(entry_ptr(pt, lastp.g)[0]);
                    const u128 lastk = lastp.rc ? kmer_rc<K>(Fl) : Fl;
                    if (L & 1) {
                        const uint32_t mid = L / 2;                       // base index in the walk's orientation
                        if (mid < (uint32_t)K) rev = (kmer_base<K>(first, (int)mid) & 2u) != 0u;
                        else if (mid >= n - 1) rev = (kmer_base<K>(lastk, (int)(mid - (n - 1))) & 2u) != 0u;
                        else pre_left = mid - (K - 1);                    // the base appended at that step: walk there first
                    } else rev = lt128(kmer_rc<K>(lastk), first);         // rc(S) begins with rc(last k-mer)
                }
                if (!pre_left) begin_write();
            }
            continue;
        }
        if (active) {
            if (pre_left) {                                              // on the way to the middle base; nothing is marked
                const uint32_t base = one_succ(walk_ctx(p));
                WalkPos q;
                if (!walk_follow(pt, p, &q)) { atomicOr(bad, 1u); active = false; }
                else { p = q; if (--pre_left == 0) { rev = (base & 2u) != 0u; begin_write(); } }
            } else if (s < n) {
                const uint32_t base = one_succ(walk_ctx(p));
                WalkPos q;
                if (!walk_follow(pt, p, &q)) { atomicOr(bad, 1u); active = false; }
                else { mark(p, s - 1); emit(base); p = q; ++s; }
            } else {
                mark(p, n - 1);
                if (!rev && (L & 3)) out[(L - 1) >> 2] = (uint8_t)acc;   // the last, partly filled byte
                recs[eno].flags = rflags | (rev ? ER_STORED_REV : 0u);
                active = false;
            }
        }
    }
}

} // namespace dfk
