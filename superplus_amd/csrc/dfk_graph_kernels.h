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

__device__ __forceinline__ uint4* entry_ptr(const PartTable& pt, uint64_t g)
{
    uint32_t lo = 0, hi = pt.n_parts;                 // start[lo] <= g < start[hi]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pt.start[mid] <= g) lo = mid; else hi = mid; }
    return pt.ptr[lo] + 2 * (g - pt.start[lo]);
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

template <int K>
__global__ void __launch_bounds__(256)
k_graph_index(PartTable pt, uint64_t n, uint32_t* __restrict__ index, uint64_t mask)
{
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (uint64_t)gridDim.x * 256) {
        const uint4 a = *entry_ptr(pt, g);
        const uint64_t w0 = (uint64_t)a.x | ((uint64_t)a.y << 32), w1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
        uint64_t s = set_hash(w0, w1) & mask;
        while (atomicCAS(&index[s], GRAPH_EMPTY, (uint32_t)g) != GRAPH_EMPTY) s = (s + 1) & mask;
    }
}

// The entry of k-mer `v` (any orientation) and its context as seen travelling in v's orientation: EdgeBuilder::lookup
// (BuildReadQGraph48.cc:467-478).  Returns GRAPH_EMPTY if the k-mer is not in the dictionary (cannot happen for a
// neighbour named by a context bit after recomputeAdjacencies; callers treat it as "stop").
template <int K>
__device__ __forceinline__ uint32_t graph_lookup(const PartTable& pt, const uint32_t* __restrict__ index, uint64_t mask, u128 v,
                                                 uint32_t* ctx, bool* is_pal)
{
    const u128 R = kmer_rc<K>(v);
    const bool rev = lt128(R, v);
    *is_pal = eq128(R, v);
    const u128 c = rev ? R : v;
    const u128 kw = shl128(c, 128 - KTraits<K>::BITS);
    const uint64_t w0 = kw.hi, w1 = kw.lo;
    uint64_t s = set_hash(w0, w1) & mask;
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
        s = (s + 1) & mask;
    }
    return GRAPH_EMPTY;
}

// ---- classification (buildEdge, :326-336 with upstream/downstreamExtensionPossible :399-419)
// kind of an entry, kept in its pad word until the edges are placed:
enum : uint32_t { GK_INTERIOR = 0, GK_END_DOWN = 1,      // first k-mer of an edge read in its canonical orientation
                  GK_END_UP = 2,                         // last k-mer of an edge read in its canonical orientation: walked as its reverse complement
                  GK_SINGLE = 3 };                       // an edge of one k-mer: a palindrome, or no extension either way

template <int K>
__global__ void __launch_bounds__(256)
k_graph_classify(PartTable pt, uint64_t n, const uint32_t* __restrict__ index, uint64_t mask, unsigned long long* __restrict__ n_ends)
{
    unsigned long long mine = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (uint64_t)gridDim.x * 256) {
        uint4* e = entry_ptr(pt, g);
        const u128 F = kmer_of_entry<K>(e[0]);
        uint4 b = e[1];
        const uint32_t ctx = b.y >> 24;
        uint32_t kind;
        if (eq128(kmer_rc<K>(F), F)) kind = GK_SINGLE;
        else {
            bool up = false, down = false, pal;
            uint32_t c2;
            if (n_pred(ctx) == 1) {
                const u128 p = kmer_pred<K>(F, one_pred(ctx));
                const uint32_t g2 = graph_lookup<K>(pt, index, mask, p, &c2, &pal);
                up = !pal && g2 != GRAPH_EMPTY && n_succ(c2) == 1;
            }
            if (n_succ(ctx) == 1) {
                const u128 s = kmer_succ<K>(F, one_succ(ctx));
                const uint32_t g2 = graph_lookup<K>(pt, index, mask, s, &c2, &pal);
                down = !pal && g2 != GRAPH_EMPTY && n_pred(c2) == 1;
            }
            kind = up ? (down ? GK_INTERIOR : GK_END_UP) : (down ? GK_END_DOWN : GK_SINGLE);
        }
        b.x = 0xFFFFFFFFu;                           // edge id: null until the edge is written
        b.w = kind;
        e[1] = b;
        mine += kind != GK_INTERIOR;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_ends, mine);
}

// dense list of the entries that are edge ends (kind != interior); `want_null_interior`: the list of interior entries
// still without an edge instead (the members of branch-free cycles)
__global__ void __launch_bounds__(256)
k_graph_list(PartTable pt, uint64_t n, bool want_null_interior, uint32_t* __restrict__ list, uint64_t cap, unsigned long long* __restrict__ n_list)
{
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
            if (g < n) { const uint4 b = entry_ptr(pt, g)[1]; hit = want_null_interior ? (b.w == GK_INTERIOR && b.x == 0xFFFFFFFFu) : (b.w != GK_INTERIOR); }
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

// ---- edges
// One record per canonical edge, in the order the owners reserved them (arbitrary: buildHBVFromEdges sorts).
struct EdgeRec {
    uint32_t g_start;      // entry the owner's walk starts from
    uint32_t n;            // k-mers on the edge
    uint64_t byte_off;     // of its 2-bit bases in the edge store (byte aligned, LSB-first like a .fastb)
    uint32_t flags;        // bit 0: the walk starts from the reverse complement of the entry's k-mer; bit 1: cycle
    uint32_t pad;
};
constexpr uint32_t ER_START_RC = 1u, ER_CYCLE = 2u;

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

// Pass A: every end walks its edge (EdgeBuilder::extend, :436-456) to learn its length and its other end.  The end
// whose canonical k-mer is the smaller of the two owns the edge and reserves its record; a one-k-mer edge owns itself.
// (The reference builds an edge from whichever end its thread meets first and throws away the walk that comes out
// in REV form; owning by k-mer order makes exactly one lane write each edge.)
template <int K>
__global__ void __launch_bounds__(256)
k_graph_walk_count(PartTable pt, const uint32_t* __restrict__ index, uint64_t mask, const uint32_t* __restrict__ ends, uint64_t n_ends,
                   EdgeRec* __restrict__ recs, uint64_t rec_cap, unsigned long long* __restrict__ ctr, uint32_t max_steps,
                   unsigned int* __restrict__ bad)
{
    const uint64_t rounds = (n_ends + (uint64_t)gridDim.x * 256 - 1) / ((uint64_t)gridDim.x * 256);
    for (uint64_t r = 0; r < rounds; ++r) {                              // whole waves stay together for the reservation
        const uint64_t t = (r * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        bool owner = false; uint32_t n = 1, g = 0, flags = 0;
        if (t < n_ends) {
            g = ends[t];
            const uint4* e = entry_ptr(pt, g);
            const u128 F = kmer_of_entry<K>(e[0]);
            const uint4 b = e[1];
            const uint32_t kind = b.w;
            uint32_t ctx = b.y >> 24;
            if (kind == GK_SINGLE) owner = true;
            else {
                u128 cur = F;
                if (kind == GK_END_UP) { cur = kmer_rc<K>(F); ctx = ctx_rc(ctx); flags = ER_START_RC; }
                uint32_t last = g;
                while (n_succ(ctx) == 1 && n < max_steps) {
                    const u128 nxt = kmer_succ<K>(cur, one_succ(ctx));
                    uint32_t c2; bool pal;
                    const uint32_t g2 = graph_lookup<K>(pt, index, mask, nxt, &c2, &pal);
                    if (pal || g2 == GRAPH_EMPTY || n_pred(c2) != 1) break;
                    cur = nxt; ctx = c2; last = g2; ++n;
                }
                if (n >= max_steps) atomicOr(bad, 2u);
                // the other end's canonical k-mer against ours
                const u128 other = kmer_of_entry<K>(entry_ptr(pt, last)[0]);
                owner = last == g || lt128(F, other);
            }
        }
        uint64_t eno = 0, off = 0;
        reserve_edge(owner, ((uint64_t)n + K - 1 + 3) / 4, ctr, &eno, &off);
        if (owner && eno < rec_cap) recs[eno] = EdgeRec{g, n, off, flags, 0u};
    }
}

// Pass C: what is still without an edge lies on a cycle without branches (simpleCircle, :338-365).  Every member
// walks the cycle until it meets a k-mer smaller than itself (then it is not the one) or comes back to itself: the
// smallest canonical k-mer of the cycle owns it, and the edge starts there in that k-mer's canonical orientation
// (canonicalizeCircle, :367-392).
template <int K>
__global__ void __launch_bounds__(256)
k_graph_cycles(PartTable pt, const uint32_t* __restrict__ index, uint64_t mask, const uint32_t* __restrict__ members, uint64_t n_members,
               EdgeRec* __restrict__ recs, uint64_t rec_cap, unsigned long long* __restrict__ ctr, uint32_t max_steps,
               unsigned int* __restrict__ bad)
{
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
                const uint32_t g2 = graph_lookup<K>(pt, index, mask, nxt, &c2, &pal);
                if (g2 == GRAPH_EMPTY) { owner = false; break; }
                if (g2 == g) break;                                          // back at the start: the whole cycle seen
                if (lt128(kmer_of_entry<K>(entry_ptr(pt, g2)[0]), F)) { owner = false; break; }
                cur = nxt; ctx = c2; ++n;
            }
        }
        uint64_t eno = 0, off = 0;
        reserve_edge(owner, ((uint64_t)n + K - 1 + 3) / 4, ctr, &eno, &off);
        if (owner && eno < rec_cap) recs[eno] = EdgeRec{g, n, off, ER_CYCLE, 0u};
    }
}

// Pass B: the owner walks its edge again and writes it: the bases into the edge store, and into every k-mer's entry
// the edge number and the k-mer's offset on it (KDef::set, :488-491 -- as in the reference the offset takes the
// place of the count, kmers/ReadPather.h:122-127).  The edge is stored in canonical form (addEdge :480-486): FWD or
// palindrome as walked, REV reverse-complemented, with the offsets counted from the other end.
//   getCanonicalForm (dna/CanonicalForm.h:32-46): odd length -> REV iff the middle base is G or T; even length ->
//   outside-in against the complement of the mirror base, which the first and the last k-mer of the walk decide
//   (they differ as k-mers, so one of their K positions differs).
template <int K>
__global__ void __launch_bounds__(256)
k_graph_walk_write(PartTable pt, const uint32_t* __restrict__ index, uint64_t mask, EdgeRec* __restrict__ recs, uint64_t e_lo, uint64_t e_hi,
                   uint8_t* __restrict__ store, unsigned int* __restrict__ bad)
{
    for (uint64_t eno = e_lo + (uint64_t)blockIdx.x * 256 + threadIdx.x; eno < e_hi; eno += (uint64_t)gridDim.x * 256) {
        const EdgeRec R = recs[eno];
        const uint32_t n = R.n, L = n + K - 1;
        uint4* e0 = entry_ptr(pt, R.g_start);
        const u128 F0 = kmer_of_entry<K>(e0[0]);
        uint32_t ctx0 = e0[1].y >> 24;
        u128 first = F0;
        if (R.flags & ER_START_RC) { first = kmer_rc<K>(F0); ctx0 = ctx_rc(ctx0); }
        // ---- orientation of the stored edge
        bool rev = false;
        if (n > 1) {
            if (L & 1) {
                const uint32_t mid = L / 2;                               // base index in the walk's orientation
                uint32_t base;
                if (mid < (uint32_t)K) base = kmer_base<K>(first, (int)mid);
                else {                                                    // the base appended at step mid - (K-1)
                    u128 cur = first; uint32_t ctx = ctx0; base = 0;
                    for (uint32_t s = 1; s <= mid - (K - 1); ++s) {
                        base = one_succ(ctx);
                        cur = kmer_succ<K>(cur, base);
                        uint32_t c2; bool pal;
                        if (graph_lookup<K>(pt, index, mask, cur, &c2, &pal) == GRAPH_EMPTY) { atomicOr(bad, 1u); break; }
                        ctx = c2;
                    }
                }
                rev = (base & 2u) != 0u;
            }
        }
        // (even length: decided below, once the last k-mer is known -- the walk runs first without writing)
        u128 cur = first; uint32_t ctx = ctx0;
        if (n > 1 && !(L & 1)) {
            for (uint32_t s = 1; s < n; ++s) {
                cur = kmer_succ<K>(cur, one_succ(ctx));
                uint32_t c2; bool pal;
                if (graph_lookup<K>(pt, index, mask, cur, &c2, &pal) == GRAPH_EMPTY) { atomicOr(bad, 1u); break; }
                ctx = c2;
            }
            const u128 rl = kmer_rc<K>(cur);                             // rc(S) begins with rc(last k-mer)
            rev = lt128(rl, first);
            cur = first; ctx = ctx0;
        }
        // ---- write.  Position p of the walk's sequence is stored at q = rev ? L-1-p : p, as base or its complement.
        uint8_t* out = store + R.byte_off;
        const uint32_t n_bytes = (L + 3) / 4;
        for (uint32_t i = 0; i < n_bytes; ++i) out[i] = 0;               // (a lane owns whole bytes: edges are byte aligned)
        auto put = [&](uint32_t p, uint32_t base) {
            const uint32_t q = rev ? L - 1 - p : p, v = rev ? 3u - base : base;
            out[q >> 2] |= (uint8_t)(v << (2 * (q & 3)));
        };
        for (int i = 0; i < K; ++i) put((uint32_t)i, kmer_base<K>(first, i));
        auto mark = [&](uint4* e, uint32_t step) {                       // entry <- (edge, offset)
            uint4 b = e[1];
            const uint32_t off = rev ? n - 1 - step : step;
            b.x = (uint32_t)eno;
            b.y = (b.y & 0xFF000000u) | (off & 0xFFFFFFu);
            b.w = 0;
            e[1] = b;
        };
        mark(e0, 0);
        for (uint32_t s = 1; s < n; ++s) {
            const uint32_t base = one_succ(ctx);
            cur = kmer_succ<K>(cur, base);
            uint32_t c2; bool pal;
            const uint32_t g2 = graph_lookup<K>(pt, index, mask, cur, &c2, &pal);
            if (g2 == GRAPH_EMPTY) { atomicOr(bad, 1u); break; }
            ctx = c2;
            put(K - 1 + s, base);
            mark(entry_ptr(pt, g2), s);
        }
        recs[eno].pad = rev ? 1u : 0u;
    }
}

} // namespace dfk
