// superplus_amd/csrc/dfk_paths_kernels.h -- SURVEY 8(f)-2: read pathing on the device (gfx950, wave64).
//
// What the reference does per read in pathReads (paths/long/BuildReadQGraph48.cc:1420-1442, parallelForBatch over
// HBVPather::algorithmTwo) is done here by one lane per read: dictionary look-ups through the index the edge builder
// left behind (every entry carries (edge, offset) by then, dfk_graph_kernels.h), base comparisons against the edge store,
// and the graph rules of algorithmTwo / ExtendReadPath against device copies of the HyperBasevector's tables.  Integer
// work except one double multiply-subtract per matching base of an extension score (ExtendReadPath.cc:50).  Bound by
// the latency of dependent random reads (index slot -> entry -> edge bases), like the edge walks.
//
// Reference lines each device function follows:
//   path_parts     Pather::path                         BuildReadQGraph48.cc:685-733   (matchLen :532-541, CF<K>::isRC dna/CanonicalForm.h:84-91)
//   PartD          EdgeLoc / PathPart                   :593-682
//   conforming     PathPart::isConformingCapturedGap    :657-666
//   joinable       Pather::isJoinable                   :796-803
//   edit_parts     HBVPather::algorithmTwo              :1212-1317 (the part edits)
//   to_read_path   HBVPather::pathPartsToReadPath       :1365-1402
//   score_overlap  scoreLeftOverlap / scoreRightOverlap paths/long/ExtendReadPath.cc:15-113
//   extend_path    attemptLeftward/RightwardExtension   paths/long/ExtendReadPath.cc:130-378
#pragma once
#include "dfk_graph_kernels.h"

namespace dfk {

// A path part: `edge` = canonical edge (as numbered by the edge builder), `off` = k-mer offset on the edge in the
// orientation the read runs along it, stored as ~off when that is the reverse complement (EdgeLoc :596-598), `len` =
// k-mers covered, `elen` = k-mers on the edge; elen == 0 is a gap, of which only `len` means anything.
struct PartD { uint32_t edge; int32_t off; uint32_t len, elen; };
__device__ __forceinline__ bool part_gap(const PartD& p) { return p.elen == 0; }
__device__ __forceinline__ bool part_rc(const PartD& p) { return p.off < 0; }
__device__ __forceinline__ uint32_t part_off(const PartD& p) { return p.off < 0 ? (uint32_t)~p.off : (uint32_t)p.off; }
__device__ __forceinline__ bool part_same_edge(const PartD& a, const PartD& b) { return a.edge == b.edge && part_rc(a) == part_rc(b); }
__device__ __forceinline__ PartD part_gap_of(uint32_t n) { return PartD{0xFFFFFFFFu, 0, n, 0u}; }

// Device copy of what pathing needs of the HyperBasevector (built on the host by build_hbv, dfk_graph.inc)
struct PathGraph {
    const EdgeRec* ce;            // canonical edges: n = k-mers, byte_off = first byte of its bases in `store`
    const uint8_t* store;         // 2-bit bases, LSB-first, byte aligned per edge
    const int2* xlat;             // canonical edge -> HBV edge id {forward, reverse complement}  (fwdEdgeXlat / revEdgeXlat)
    const uint32_t* he_ce;        // HBV edge -> canonical edge << 1 | is reverse complement
    const int32_t* he_left;       // ToLeft / ToRight
    const int32_t* he_right;
    const uint32_t* from_start;   // per vertex, CSR over digraphE's own (sorted) adjacency order: From(v) / FromEdgeObj(v)
    const int32_t* from_vtx;
    const int32_t* from_edge;
    const uint32_t* to_start;     // To(v) / ToEdgeObj(v)
    const int32_t* to_vtx;
    const int32_t* to_edge;
    uint32_t n_ce, n_he, n_v;
    uint64_t store_words;         // 32-bit words of `store` (it is read as aligned words by the base comparisons)
};

__device__ __forceinline__ uint32_t seq_base(const uint8_t* __restrict__ p, uint64_t i) { return (p[i >> 2] >> (2 * (i & 3))) & 3u; }

// sixteen bases from base index `first` of a 2-bit stream that starts at byte `byte0` of the word array `words` (base `first`
// in bits 0-1): two aligned loads and a funnel shift.  `n_words` bounds the second load (the stream's last word has no successor).
__device__ __forceinline__ uint32_t bases16(const uint32_t* __restrict__ words, uint64_t n_words, uint64_t byte0, uint64_t first)
{
    const uint64_t bit = 8 * byte0 + 2 * first, wi = bit >> 5;
    const uint32_t lo = words[wi], hi = wi + 1 < n_words ? words[wi + 1] : 0u;
    return alignbit(hi, lo, (uint32_t)bit & 31u);
}
// the order of the sixteen 2-bit fields of x reversed
__device__ __forceinline__ uint32_t rev2_32(uint32_t x) { x = __brev(x); return ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1); }

// a canonical edge read along (rc = false) or against (rc = true) its stored orientation
struct EdgeView {
    const uint8_t* p; uint32_t L; bool rc;
    __device__ __forceinline__ uint32_t at(uint32_t i) const { return rc ? 3u - seq_base(p, L - 1u - i) : seq_base(p, i); }
};
template <int K> __device__ __forceinline__ EdgeView edge_view(const PathGraph& G, uint32_t c, bool rc)
{ const EdgeRec r = G.ce[c]; return EdgeView{G.store + r.byte_off, r.n + (uint32_t)K - 1u, rc}; }
template <int K> __device__ __forceinline__ EdgeView hbv_edge_view(const PathGraph& G, int32_t e)
{ const uint32_t x = G.he_ce[e]; return edge_view<K>(G, x >> 1, (x & 1u) != 0u); }
__device__ __forceinline__ int32_t hbv_kmers(const PathGraph& G, int32_t e) { return (int32_t)G.ce[G.he_ce[e] >> 1].n; }
__device__ __forceinline__ uint32_t to_size(const PathGraph& G, int32_t v) { return G.to_start[v + 1] - G.to_start[v]; }
__device__ __forceinline__ uint32_t from_size(const PathGraph& G, int32_t v) { return G.from_start[v + 1] - G.from_start[v]; }
__device__ __forceinline__ int32_t part_hbv_edge(const PathGraph& G, const PartD& p) { const int2 x = G.xlat[p.edge]; return part_rc(p) ? x.y : x.x; }

// ---- Pather::path
// KmerDict::findEntry (kmers/ReadPather.h:222-225) through the edge builder's index: entry index or GRAPH_EMPTY, and the
// entry's second half ((edge, offset) since the graph was built) read beside its key rather than after it.
struct DictKey { uint64_t w0, w1, slot; };
template <int K> __device__ __forceinline__ DictKey dict_key(uint64_t n_slots, u128 v)
{
    const u128 R = kmer_rc<K>(v);
    const u128 c = lt128(R, v) ? R : v;
    const u128 kw = shl128(c, 128 - KTraits<K>::BITS);
    return DictKey{kw.hi, kw.lo, index_home(set_hash(kw.hi, kw.lo), n_slots)};
}
__device__ __forceinline__ bool key_is(const uint4 a, const DictKey& k)
{ return ((uint64_t)a.x | ((uint64_t)a.y << 32)) == k.w0 && ((uint64_t)a.z | ((uint64_t)a.w << 32)) == k.w1; }
// the probe sequence from slot s on
__device__ __forceinline__ uint32_t dict_probe(const PartLds& pt, const uint32_t* __restrict__ index, uint64_t n_slots, const DictKey& k, uint64_t s, uint4* second)
{
    for (uint32_t guard = 0; guard < 1u << 20; ++guard) {
        const uint32_t g = index[s];
        if (g == GRAPH_EMPTY) return GRAPH_EMPTY;
        const uint4* e = entry_ptr(pt, g);
        const uint4 a = e[0], b = e[1];
        if (key_is(a, k)) { *second = b; return g; }
        s = index_next(s, n_slots);
    }
    return GRAPH_EMPTY;
}
// In front of the index, a filter of two bytes per k-mer: one 32-bit word per key, four of its bits.  A wrong base in a read
// makes K look-ups in a row miss, and a miss in the index is 2.7 scattered reads on average (slots at load 1/2, and the key
// of every occupied slot on the way lies in its entry); the filter answers 99.5 % of them with one.  What it lets through
// the index decides as before.
struct KmerFilter { const uint32_t* words; uint64_t n_words; };
__device__ __forceinline__ void filter_place(const DictKey& k, uint64_t n_words, uint64_t* word, uint32_t* bits)
{
    const uint64_t h = (k.w0 ^ (k.w1 * 0x9E3779B97F4A7C15ull)) * 0xD6E8FEB86659FD93ull;
    const uint64_t h2 = (h ^ (h >> 32)) * 0xD6E8FEB86659FD93ull;
    *word = __umul64hi(h2, n_words);
    const uint32_t x = (uint32_t)(h >> 7);
    *bits = (1u << (x & 31)) | (1u << ((x >> 5) & 31)) | (1u << ((x >> 10) & 31)) | (1u << ((x >> 15) & 31));
}
template <int K>
__global__ void __launch_bounds__(256)
k_filter_build(PartTable pt_arg, uint64_t n, uint32_t* __restrict__ words, uint64_t n_words)
{
    __shared__ PartLds pt;
    part_lds_init(pt_arg, pt);
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (uint64_t)gridDim.x * 256) {
        const uint4 a = entry_ptr(pt, g)[0];
        const DictKey k{(uint64_t)a.x | ((uint64_t)a.y << 32), (uint64_t)a.z | ((uint64_t)a.w << 32), 0};
        uint64_t w; uint32_t bits;
        filter_place(k, n_words, &w, &bits);
        atomicOr(&words[w], bits);
    }
}
template <int K>
__device__ __forceinline__ uint32_t dict_find(const PartLds& pt, const uint32_t* __restrict__ index, uint64_t n_slots, const KmerFilter& F, u128 v, uint4* second)
{
    const DictKey k = dict_key<K>(n_slots, v);
    if (F.words) {
        uint64_t w; uint32_t bits;
        filter_place(k, F.n_words, &w, &bits);
        if ((F.words[w] & bits) != bits) return GRAPH_EMPTY;
    }
    return dict_probe(pt, index, n_slots, k, k.slot, second);
}

constexpr uint32_t PARTS_OVERFLOW = 0xFFFFFFFFu;
template <int K>
__device__ uint32_t path_parts(const PartLds& pt, const uint32_t* __restrict__ index, uint64_t n_slots, const KmerFilter& F, const PathGraph& G,
                               const uint32_t* __restrict__ rwords, uint64_t r_nwords, uint64_t r_byte0,
                               const uint8_t* __restrict__ read, uint32_t n, PartD* __restrict__ parts, uint32_t room, unsigned int* __restrict__ bad)
{
    // (`room` parts fit; PARTS_OVERFLOW if the read wants more -- with room for one part per k-mer position it cannot)
    if (n < (uint32_t)K) { parts[0] = part_gap_of(n); return 1; }
    const uint32_t* ewords = reinterpret_cast<const uint32_t*>(G.store);
    uint32_t np = 0, at = 0;
    const uint32_t stop = n - K + 1;
    while (at != stop) {
        // the k-mer at `at`: its 2K bits straight from the stream (sixteen bases a word), then turned big-endian as KMer keeps them
        u128 ks;
        {
            const uint32_t w0 = bases16(rwords, r_nwords, r_byte0, at), w1 = bases16(rwords, r_nwords, r_byte0, at + 16u),
                           w2 = bases16(rwords, r_nwords, r_byte0, at + 32u), w3 = K > 48 ? bases16(rwords, r_nwords, r_byte0, at + 48u) : 0u;
            ks = u128{(uint64_t)w0 | ((uint64_t)w1 << 32), (uint64_t)w2 | ((uint64_t)w3 << 32)};
            const u128 m = KTraits<K>::mask();
            ks.lo &= m.lo; ks.hi &= m.hi;
        }
        u128 km = shr128(u128{rev2_64(ks.hi), rev2_64(ks.lo)}, 128 - KTraits<K>::BITS);
        uint4 b;
        // (a read's first k-mer is in the dictionary more often than not: no filter in front of that look-up)
        uint32_t hit = dict_find<K>(pt, index, n_slots, np ? F : KmerFilter{nullptr, 0}, km, &b);
        if (hit == GRAPH_EMPTY) {
            uint32_t missed = 1, nxt = at + K;
            ++at;
            while (nxt != n) {
                km = kmer_succ<K>(km, seq_base(read, nxt)); ++nxt;
                if ((hit = dict_find<K>(pt, index, n_slots, F, km, &b)) != GRAPH_EMPTY) break;
                ++missed; ++at;
            }
            if (np == room) return PARTS_OVERFLOW;
            parts[np++] = part_gap_of(missed);
        }
        if (hit != GRAPH_EMPTY) {
            if (np == room) return PARTS_OVERFLOW;
            const uint32_t c = b.x;
            if (c >= G.n_ce) { atomicOr(bad, 4u); parts[np++] = part_gap_of(stop - at); break; }   // (an entry without an edge: the graph is not the dictionary's)
            int32_t off = (int32_t)(b.y & 0xFFFFFFu);
            const EdgeRec er = G.ce[c];
            const EdgeView fw{G.store + er.byte_off, er.n + (uint32_t)K - 1u, false};
            // CF<K>::isRC: at the first position where the read's k-mer differs from its own reverse complement, does it
            // differ from the edge's k-mer too?  (a palindrome never does)
            const u128 R = kmer_rc<K>(km);
            bool rc = false;
            if (!eq128(R, km)) {
                const u128 x{km.lo ^ R.lo, km.hi ^ R.hi};
                const int lead = x.hi ? __clzll((long long)x.hi) : 64 + __clzll((long long)x.lo);     // leading equal bits of the 128-bit words
                const int i = (lead - (128 - KTraits<K>::BITS)) >> 1;
                rc = kmer_base<K>(km, i) != fw.at((uint32_t)off + (uint32_t)i);
            }
            // matchLen (:532-541), sixteen bases a step while both sequences have them, then base by base
            uint32_t len = 1;
            if (!rc) {
                uint32_t i = at + K, j = (uint32_t)off + K;
                for (;;) {
                    if (i + 16u > n || j + 16u > fw.L) break;
                    const uint32_t x = bases16(rwords, r_nwords, r_byte0, i) ^ bases16(ewords, G.store_words, er.byte_off, j);
                    if (x) { const uint32_t m = (uint32_t)__builtin_ctz(x) >> 1; len += m; i = n; break; }   // (i = n: the loop below has nothing left to do)
                    len += 16u; i += 16u; j += 16u;
                }
                for (; i < n && j < fw.L && seq_base(read, i) == fw.at(j); ++i, ++j) ++len;
            } else {
                const EdgeView bw{fw.p, fw.L, true};
                off = (int32_t)fw.L - off;
                uint32_t i = at + K, j = (uint32_t)off;                 // j: index on the reverse complement = stored position L-1-j, walked downwards
                for (;;) {
                    if (i + 16u > n || j + 16u > bw.L) break;
                    const uint32_t e = bases16(ewords, G.store_words, er.byte_off, bw.L - 16u - j);   // stored positions L-16-j .. L-1-j
                    const uint32_t x = bases16(rwords, r_nwords, r_byte0, i) ^ rev2_32(~e);
                    if (x) { const uint32_t m = (uint32_t)__builtin_ctz(x) >> 1; len += m; i = n; break; }
                    len += 16u; i += 16u; j += 16u;
                }
                for (; i < n && j < bw.L && seq_base(read, i) == bw.at(j); ++i, ++j) ++len;
                off -= K;
            }
            parts[np++] = PartD{c, rc ? ~off : off, len, fw.L - (uint32_t)K + 1u};
            at += len;
        }
    }
    return np;
}

// ---- the part edits of algorithmTwo
__device__ __forceinline__ bool conforming(const PartD& before, const PartD& gap, const PartD& after)
{
    uint32_t dist = part_off(after) - (part_off(before) + before.len);         // all unsigned, as in the reference
    if (!part_same_edge(before, after)) dist += before.elen;
    const int32_t d = (int32_t)(gap.len - dist);
    return (uint32_t)(d < 0 ? -d : d) <= 3u;                                    // HBVPather::MAX_JITTER
}

template <int K>
__device__ bool joinable(const PathGraph& G, const PartD& a, const PartD& b)
{
    if (a.edge == b.edge) return true;
    const EdgeView e1 = edge_view<K>(G, a.edge, part_rc(a)), e2 = edge_view<K>(G, b.edge, part_rc(b));
    for (uint32_t i = 0; i < (uint32_t)K - 1u; ++i)
        if (e1.at(e1.L - (K - 1) + i) != e2.at(i)) return false;
    return true;
}

template <int K>
__device__ uint32_t edit_parts(const PathGraph& G, PartD* __restrict__ parts, uint32_t np)
{
    // seeds on short hanging edges become gaps; neighbouring gaps merge (in place: the write index never passes the read index)
    uint32_t w = 0;
    for (uint32_t i = 0; i < np; ++i) {
        PartD p = parts[i];
        if (!part_gap(p)) {
            const int32_t e = part_hbv_edge(G, p), vl = G.he_left[e], vr = G.he_right[e];
            if (to_size(G, vl) == 0 && to_size(G, vr) > 1 && from_size(G, vr) > 0 && p.elen <= 100u) p = part_gap_of(p.len);
        }
        if (part_gap(p) && w && part_gap(parts[w - 1])) parts[w - 1].len += p.len;
        else parts[w++] = p;
    }
    np = w;
    // the first captured gap the graph does not explain ends the path
    if (np >= 3) {
        uint32_t seeds = part_gap(parts[0]) ? 0u : 1u;
        for (uint32_t i = 1; i + 1 < np; ++i) {
            if (!part_gap(parts[i])) { ++seeds; continue; }
            if (conforming(parts[i - 1], parts[i], parts[i + 1]) && joinable<K>(G, parts[i - 1], parts[i + 1])) continue;
            if (seeds > 1) {
                uint32_t tail = parts[i - 1].len;
                for (uint32_t j = i; j < np; ++j) tail += parts[j].len;
                parts[i - 1] = part_gap_of(tail);
                np = i;
            } else {
                uint32_t more = 0;
                for (uint32_t j = i + 1; j < np; ++j) more += parts[j].len;
                parts[i].len += more;
                np = i + 1;
            }
            break;
        }
    }
    // a last seed of at most five k-mers at the very start of its edge is not trusted.  (The part in front of a final gap
    // can itself be a gap -- the cut above leaves one behind a gap -- and reads as offset 0 like the reference's.)
    if (part_gap(parts[np - 1]) && np > 1) {
        const PartD seed = parts[np - 2];
        if (part_off(seed) == 0 && seed.len <= 5u) { parts[np - 2] = part_gap_of(parts[np - 1].len + seed.len); --np; }
    } else if (!part_gap(parts[np - 1])) {
        const PartD seed = parts[np - 1];
        if (part_off(seed) == 0 && seed.len <= 5u) parts[np - 1] = part_gap_of(seed.len);
    }
    return np;
}

// ---- ExtendReadPath
// Mismatches cost the base's quality (2 counts as 20) plus a running penalty that every matching base shrinks by a
// fifth -- `penalty -= pDecay*penalty` on an unsigned with a double on the right: the product and the difference are IEEE
// doubles, rounded separately (no fused multiply-add), and the result is truncated.  Read bases left over past the edge
// cost 10 each.
__device__ __forceinline__ uint32_t decay_penalty(uint32_t penalty)
{
#pragma clang fp contract(off)
    const double p = (double)penalty;
    const double d = 0.2 * p;
    return (uint32_t)(p - d);
}

__device__ uint32_t score_overlap(const uint8_t* __restrict__ read, const uint8_t* __restrict__ q, uint32_t n, uint32_t start,
                                  const EdgeView& edge, uint32_t K, bool left)
{
    uint32_t sum = 0, penalty = 0;
    int64_t r = left ? (int64_t)start - 1 : (int64_t)n - (int64_t)start;          // read index, moving outwards
    int64_t e = left ? (int64_t)edge.L - (int64_t)K : (int64_t)K - 1;             // edge index beside the shared K-1 bases
    const int64_t step = left ? -1 : 1;
    while (r >= 0 && r < (int64_t)n && e >= 0 && e < (int64_t)edge.L) {
        if (seq_base(read, (uint64_t)r) != edge.at((uint32_t)e)) { const uint32_t qv = q[r]; penalty += qv == 2u ? 20u : qv; sum += penalty; }
        else if (penalty > 0) penalty = decay_penalty(penalty);
        r += step; e += step;
    }
    while (r >= 0 && r < (int64_t)n) { sum += 10u; r += step; }
    return sum;
}

// PQVecEncoder::decode (feudal/PQVec.cc:129-188) of one read into `out` (n bytes); false if the stream does not hold n values
__device__ bool decode_quals(const uint8_t* __restrict__ pq, uint64_t p, uint64_t end, uint8_t* __restrict__ out, uint32_t n)
{
    uint32_t idx = 0;
    while (p < end) {
        const uint32_t nQs = pq[p];
        if (!nQs) break;
        if (p + 3 > end) return false;
        const uint32_t hdr = pq[p + 1] | ((uint32_t)pq[p + 2] << 8);
        const uint32_t nBits = hdr & 7u, minQ = (hdr >> 3) & 63u;
        const uint64_t blk = ((uint64_t)nQs * nBits + 24) >> 3;
        if (p + blk > end || idx + nQs > n) return false;
        uint64_t bit = 8 * (p + 1) + 9;
        for (uint32_t i = 0; i < nQs; ++i) {
            uint32_t v = 0;
            for (uint32_t k = 0; k < nBits; ++k, ++bit) v |= ((uint32_t)(pq[bit >> 3] >> (bit & 7)) & 1u) << k;
            out[idx++] = (uint8_t)(minQ + v);
        }
        p += blk;
    }
    return idx == n;
}

// ---- the quality histogram of DF's side files (10X/DF.cc:50-68 through DfTools.cc:172-238): how often quality q stands at
// position pos of a first / second read.  Kept as a DIFFERENCE table [parity][q][pos]: a run of equal qualities is +1 where it
// begins and -1 behind its end (a block of single qualities is cut into its runs), summed along pos on the host afterwards.
// Every workgroup has the part q < 64, pos <= 256 of it in LDS -- 1.8e9 reads put their one or two runs on a handful of
// addresses, which global atomics would take one after the other -- and adds it to the global table at the end; what lies
// outside goes there directly.  Positions >= max_len are left out, as the reference leaves them.
constexpr uint32_t QH_POS = 256, QH_Q = 64;
__global__ void __launch_bounds__(1024)
k_qual_hist(const uint8_t* __restrict__ pq, const uint64_t* __restrict__ pq_off, uint64_t n_reads, uint32_t max_len,
            long long* __restrict__ diff /* [2][256][max_len + 1] */)
{
    extern __shared__ int qh_lds[];                                       // [2][QH_Q][QH_POS + 1]
    for (uint32_t i = threadIdx.x; i < 2 * QH_Q * (QH_POS + 1); i += blockDim.x) qh_lds[i] = 0;
    __syncthreads();
    auto run = [&](uint32_t par, uint32_t q, uint32_t from, uint32_t to) {      // positions [from, to)
        const uint32_t a = from < max_len ? from : max_len, b = to < max_len ? to : max_len;
        if (a == b) return;
        if (q < QH_Q && b <= QH_POS) {
            int* row = qh_lds + (par * QH_Q + q) * (QH_POS + 1);
            atomicAdd(&row[a], 1); atomicAdd(&row[b], -1);
        } else {
            long long* row = diff + ((uint64_t)par * 256 + q) * (max_len + 1);
            atomicAdd((unsigned long long*)&row[a], 1ull); atomicAdd((unsigned long long*)&row[b], ~0ull);
        }
    };
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t p = pq_off[r]; const uint64_t end = pq_off[r + 1];
        const uint32_t par = (uint32_t)(r & 1);
        uint32_t pos = 0;
        while (p < end) {
            const uint32_t nQs = pq[p];
            if (!nQs || p + 3 > end) break;
            const uint32_t hdr = pq[p + 1] | ((uint32_t)pq[p + 2] << 8), nBits = hdr & 7u, minQ = (hdr >> 3) & 63u;
            const uint64_t blk = ((uint64_t)nQs * nBits + 24) >> 3;
            if (p + blk > end) break;
            if (!nBits) run(par, minQ, pos, pos + nQs);
            else {
                uint64_t bit = 8 * (p + 1) + 9;
                uint32_t q0 = 0, from = pos;
                for (uint32_t i = 0; i < nQs; ++i, bit += nBits) {
                    const uint64_t by = bit >> 3;
                    const uint32_t w = pq[by] | (by + 1 < p + blk ? (uint32_t)pq[by + 1] << 8 : 0u);
                    const uint32_t q = minQ + ((w >> (bit & 7)) & ((1u << nBits) - 1u));
                    if (i && q != q0) { run(par, q0, from, pos + i); from = pos + i; }
                    q0 = q;
                }
                run(par, q0, from, pos + nQs);
            }
            pos += nQs; p += blk;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 2 * QH_Q * (QH_POS + 1); i += blockDim.x) {
        const int v = qh_lds[i];
        if (!v) continue;
        const uint32_t at = i % (QH_POS + 1), q = (i / (QH_POS + 1)) % QH_Q, par = i / ((QH_POS + 1) * QH_Q);
        if (at <= max_len) atomicAdd((unsigned long long*)&diff[((uint64_t)par * 256 + q) * (max_len + 1) + at], (unsigned long long)(long long)v);
    }
}

// One attempt to extend the path by one edge.  `path` points at the first edge of the path (room in front and behind is
// the caller's business); returns true when an edge was added: left = true -> *first moved down by one and *offset grown.
template <int K>
__device__ bool extend_path(const PathGraph& G, int32_t* __restrict__ path, int32_t* first, int32_t* count, int32_t* offset,
                            const uint8_t* __restrict__ read, const uint8_t* __restrict__ q, uint32_t n, bool left)
{
    if (*count == 0) return false;
    int64_t hang;
    if (left) {
        if (*offset >= 0) return false;
        hang = -(int64_t)*offset;
    } else {
        int32_t h = (int32_t)n + *offset;
        for (int32_t i = 0; i < *count; ++i) h -= hbv_kmers(G, path[*first + i]);
        h -= K - 1;
        hang = h;
    }
    if (hang < 10) return false;
    const int32_t v = left ? G.he_left[path[*first]] : G.he_right[path[*first + *count - 1]];
    const uint32_t c0 = left ? G.to_start[v] : G.from_start[v], c1 = left ? G.to_start[v + 1] : G.from_start[v + 1];
    const int32_t* __restrict__ cand = left ? G.to_edge : G.from_edge;
    const int32_t* __restrict__ far = left ? G.to_vtx : G.from_vtx;
    const uint32_t nc = c1 - c0;
    auto dead_end = [&](int32_t w) { return left ? (to_size(G, w) == 0 && from_size(G, w) == 1) : (from_size(G, w) == 0 && to_size(G, w) == 1); };
    // short edges (they cannot take the whole overhang) that do not hang are followed only when there is nothing longer,
    // they all lead to one vertex and that vertex has one way on
    if (nc != 1) {
        uint32_t n_reach = 0, n_short = 0;
        int32_t short_to = -1;
        bool one_dest = true;
        for (uint32_t i = c0; i < c1; ++i) {
            const bool hanging = dead_end(far[i]);
            const bool reaches = (int64_t)hbv_kmers(G, cand[i]) >= hang;             // edge bases - (K-1) >= overhang
            n_reach += reaches;
            if (!reaches && !hanging) { if (n_short && far[i] != short_to) one_dest = false; short_to = far[i]; ++n_short; }
        }
        if (n_short) {
            if (n_reach) return false;
            if (!one_dest) return false;
            if ((left ? to_size(G, short_to) : from_size(G, short_to)) != 1u) return false;
        }
    }
    int32_t best = -1;
    uint32_t least = 0xFFFFFFFFu;
    for (uint32_t i = c0; i < c1; ++i)
        if (nc == 1 || !dead_end(far[i])) {
            const uint32_t s = score_overlap(read, q, n, (uint32_t)hang, hbv_edge_view<K>(G, cand[i]), (uint32_t)K, left);
            if (s < least) { least = s; best = cand[i]; }
        }
    if (best == -1 || (uint64_t)least > (uint64_t)hang * 10u) return false;
    if (left) { --*first; path[*first] = best; *offset += hbv_kmers(G, best); }
    else path[*first + *count] = best;
    ++*count;
    return true;
}

// Slots: a read of L bases has s = max(1, L-K+1) slots; its parts (<= s), its path (room for 2s+2 edge ids: the path of the
// parts starts in the middle, leftward extensions grow down, rightward ones up; a path never holds more than s edges) and
// its decoded qualities (s+K >= L bytes) live in per-batch scratch arrays addressed by the exclusive scan of the slots.
__global__ void __launch_bounds__(256)
k_path_slots(const uint32_t* __restrict__ read_len, uint64_t r0, uint64_t nb, uint32_t K, uint32_t cap, uint64_t* __restrict__ slots)
{
    // low half: the parts (and path edges) a read gets room for -- one per k-mer position covers every case, `cap` nearly every
    // read at a fraction of the room; high half: its bases (the decoded qualities).  One scan gives both offsets: the host
    // sizes a batch from the longest read so that neither sum reaches 2^32 (paths_build_typed, nb_bound).
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * 256) {
        const uint32_t L = read_len[r0 + i], s = L >= K ? L - K + 1 : 1;
        slots[i] = (uint64_t)(s < cap ? s : cap) | ((uint64_t)((L + 3u) & ~3u) << 32);
    }
}

template <int K>
__global__ void __launch_bounds__(256)
k_path_reads(PartTable pt_arg, const uint32_t* __restrict__ index, uint64_t n_slots, KmerFilter F, PathGraph G,
             const uint8_t* __restrict__ packed, uint64_t packed_bytes, const uint64_t* __restrict__ base_off, const uint32_t* __restrict__ read_len,
             const uint8_t* __restrict__ pq, const uint64_t* __restrict__ pq_off, uint64_t r0, uint64_t nb,
             const uint64_t* __restrict__ slot_off, PartD* __restrict__ parts_all, int32_t* __restrict__ path_all, uint8_t* __restrict__ qual_all,
             int32_t* __restrict__ out_offset, uint32_t* __restrict__ out_len, uint32_t* __restrict__ out_first,
             unsigned long long* __restrict__ stats /* [0] reads placed, [1] path edges */, unsigned int* __restrict__ bad)
{
    __shared__ PartLds pt;
    part_lds_init(pt_arg, pt);
    unsigned long long placed = 0, n_edges = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * 256) {
        const uint64_t r = r0 + i, so2 = slot_off[i];
        const uint32_t so = (uint32_t)so2, n = read_len[r], s = (uint32_t)slot_off[i + 1] - so, s_full = n >= (uint32_t)K ? n - K + 1 : 1;
        const uint8_t* read = packed + base_off[r];
        PartD* parts = parts_all + so;
        int32_t* path = path_all + 2 * (uint64_t)so + 2 * i;
        uint8_t* q = qual_all + (so2 >> 32);
        // out of room: with one slot per k-mer position that cannot happen (every part covers a position of its own); with
        // fewer, the host does the batch again with all of them
        const unsigned int no_room = s < s_full ? 32u : 8u;
        uint32_t np = path_parts<K>(pt, index, n_slots, F, G, reinterpret_cast<const uint32_t*>(packed), (packed_bytes + 3) >> 2, base_off[r], read, n, parts, s, bad);
        if (np == PARTS_OVERFLOW) { atomicOr(bad, no_room); out_offset[i] = 0; out_len[i] = 0; out_first[i] = 0; continue; }
        np = edit_parts<K>(G, parts, np);
        // pathPartsToReadPath
        int32_t first = (int32_t)s + 1, count = 0, offset = 0;
        {
            int32_t last = -1;
            for (uint32_t j = 0; j < np; ++j) {
                const PartD p = parts[j];
                if (part_gap(p) || (last >= 0 && part_same_edge(parts[last], p))) continue;
                path[first + count++] = part_hbv_edge(G, p); last = (int32_t)j;
            }
            if (count) offset = !part_gap(parts[0]) ? (int32_t)part_off(parts[0]) : (int32_t)part_off(parts[1]) - (int32_t)parts[0].len;
        }
        // consecutive edges must meet at a vertex
        for (int32_t j = 0; j + 1 < count; ++j)
            if (G.he_right[path[first + j]] != G.he_left[path[first + j + 1]]) { count = j + 1; break; }
        if (count) {
            // the qualities are needed by the extension scores only: decoded once an extension can be attempted at all
            bool have_q = false;
            for (int side = 0; side < 2; ++side) {
                const bool left = side == 0;
                for (;;) {
                    int32_t h;
                    if (left) h = -offset;
                    else { h = (int32_t)n + offset - (K - 1); for (int32_t j = 0; j < count; ++j) h -= hbv_kmers(G, path[first + j]); }
                    if (h < 10) break;
                    // (a path never holds more edges than the read has k-mers: every edge of it covers a k-mer position of
                    // its own.  That bounds the scratch; an overhang of ten k-mers beside a full path would disprove it.)
                    if (count >= (int32_t)s) { atomicOr(bad, no_room); break; }
                    if (!have_q) {
                        have_q = true;
                        if (!decode_quals(pq, pq_off[r], pq_off[r + 1], q, n)) { atomicOr(bad, 16u); side = 2; break; }
                    }
                    if (!extend_path<K>(G, path, &first, &count, &offset, read, q, n, left)) break;
                }
            }
        }
        out_offset[i] = offset; out_len[i] = (uint32_t)count; out_first[i] = (uint32_t)first;
        placed += count != 0; n_edges += (unsigned long long)count;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { placed += __shfl_down(placed, d, 64); n_edges += __shfl_down(n_edges, d, 64); }
    // (one pair of atomics a block, not a wave: the two words are the same for the whole grid, and the kernel runs once a batch)
    __shared__ unsigned long long tally[2][4];
    if ((threadIdx.x & 63) == 0) { tally[0][threadIdx.x >> 6] = placed; tally[1][threadIdx.x >> 6] = n_edges; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long p = tally[0][0] + tally[0][1] + tally[0][2] + tally[0][3], e = tally[1][0] + tally[1][1] + tally[1][2] + tally[1][3];
        if (p | e) { atomicAdd(&stats[0], p); atomicAdd(&stats[1], e); }
    }
}

// a.paths element sizes (ReadPath::writeFeudal, paths/long/ReadPath.h:56-58: i32 offset, u32 lastSkip, the edge ids)
__global__ void __launch_bounds__(256)
k_path_sizes(const uint32_t* __restrict__ out_len, uint64_t nb, uint64_t* __restrict__ sizes)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * 256) sizes[i] = 8ull + 4ull * out_len[i];
}

// the batch's share of the file's variable-length data, and every element's absolute file offset
__global__ void __launch_bounds__(256)
k_path_emit(const uint64_t* __restrict__ size_off, const uint64_t* __restrict__ slot_off, const int32_t* __restrict__ path_all,
            const int32_t* __restrict__ out_offset, const uint32_t* __restrict__ out_len, const uint32_t* __restrict__ out_first,
            uint64_t nb, uint32_t* __restrict__ var, uint32_t* __restrict__ elem_off /* byte offset of every element inside the batch's data */)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * 256) {
        const uint64_t at = size_off[i];
        elem_off[i] = (uint32_t)at;
        uint32_t* o = var + (at >> 2);
        o[0] = (uint32_t)out_offset[i]; o[1] = 0u;                               // mLastSkip: never set (ReadPath.h:27-29)
        const int32_t* p = path_all + 2 * (uint64_t)(uint32_t)slot_off[i] + 2 * i + out_first[i];
        for (uint32_t j = 0; j < out_len[i]; ++j) o[2 + j] = (uint32_t)p[j];
    }
}

// ============================================================================ row f-4: paths index, duplicate marks
// What MarkDups needs of a read besides its path, gathered while the reads are at hand (dfk_paths_build): the first five
// bases as a base-4 number (10X/SecretOps.cc:430-433) in bits 0-9, the sum of its qualities (:474-481) above them.
__global__ void __launch_bounds__(256)
k_read_digest(const uint8_t* __restrict__ packed, const uint64_t* __restrict__ base_off, const uint32_t* __restrict__ read_len,
              const uint8_t* __restrict__ pq, const uint64_t* __restrict__ pq_off, uint64_t pq_nbytes, uint64_t n, uint32_t* __restrict__ digest)
{
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (uint64_t)gridDim.x * 256) {
        const uint8_t* read = packed + base_off[r];
        const uint32_t L = read_len[r];
        uint32_t head = 0;
        for (uint32_t j = 0; j < 5; ++j) head = 4u * head + (j < L ? seq_base(read, j) : 0u);
        uint32_t sum = 0;
        uint64_t p = pq_off[r];
        const uint64_t end = pq_off[r + 1];
        while (p < end) {                                                 // PQVec blocks (feudal/PQVec.cc:87-127)
            const uint32_t nQs = pq[p];
            if (!nQs || p + 3 > end) break;
            const uint32_t hdr = pq[p + 1] | ((uint32_t)pq[p + 2] << 8);
            const uint32_t nBits = hdr & 7u, minQ = (hdr >> 3) & 63u;
            const uint64_t blk = ((uint64_t)nQs * nBits + 24) >> 3;
            if (p + blk > end) break;
            sum += nQs * minQ;
            if (nBits) {
                uint64_t bit = 8 * (p + 1) + 9;
                const uint32_t fmask = (1u << nBits) - 1u;
                for (uint32_t i = 0; i < nQs; ++i, bit += nBits) {
                    const uint64_t by = bit >> 3;                              // (a field of <= 7 bits spans at most two bytes)
                    const uint32_t w = pq[by] | (by + 1 < pq_nbytes ? (uint32_t)pq[by + 1] << 8 : 0u);
                    sum += (w >> (bit & 7)) & fmask;
                }
            }
            p += blk;
        }
        digest[r] = head | (min(sum, 0x3FFFFFu) << 10);
    }
}

// ---- writePathsIndex (10X/PathsIndex.cc:23-146): the (edge, read) pairs of every path entry, in read order
__global__ void __launch_bounds__(256)
k_pidx_pairs(const uint32_t* __restrict__ var, const uint32_t* __restrict__ elem_off, uint64_t nb, uint64_t r0, uint64_t var_bytes,
             uint64_t pair_base, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ counts)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * 256) {
        const uint64_t w0 = elem_off[i] >> 2, w1 = (i + 1 < nb ? (uint64_t)elem_off[i + 1] : var_bytes) >> 2;
        // the element's first path entry is entry number (w0 - 2 i) of the batch: every element before it spent two words on offset and lastSkip
        uint64_t at = pair_base + (w0 - 2 * i);
        for (uint64_t w = w0 + 2; w < w1; ++w, ++at) { const uint32_t e = var[w]; keys[at] = e; vals[at] = (uint32_t)(r0 + i); atomicAdd(&counts[e], 1u); }
    }
}

// the same per edge RANGE [e0, e1) (the index is built range by range when the whole would not fit, or holds 2^32 entries or
// more): how many of a read's entries lie in the range, then -- at the exclusive scan of those counts, so that the reads stay
// in order -- the pairs themselves, keyed by edge - e0
__global__ void __launch_bounds__(256)
k_pidx_count(const uint32_t* __restrict__ var, const uint32_t* __restrict__ elem_off, uint64_t nb, uint64_t var_bytes, uint32_t* __restrict__ counts)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * 256) {
        const uint64_t w0 = elem_off[i] >> 2, w1 = (i + 1 < nb ? (uint64_t)elem_off[i + 1] : var_bytes) >> 2;
        for (uint64_t w = w0 + 2; w < w1; ++w) atomicAdd(&counts[var[w]], 1u);
    }
}
__global__ void __launch_bounds__(256)
k_pidx_range_count(const uint32_t* __restrict__ var, const uint32_t* __restrict__ elem_off, uint64_t nb, uint64_t var_bytes, uint32_t e0, uint32_t e1, uint64_t* __restrict__ n_in /* [nb + 1] */)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i <= nb; i += (uint64_t)gridDim.x * 256) {
        uint64_t n = 0;
        if (i < nb) {
            const uint64_t w0 = elem_off[i] >> 2, w1 = (i + 1 < nb ? (uint64_t)elem_off[i + 1] : var_bytes) >> 2;
            for (uint64_t w = w0 + 2; w < w1; ++w) { const uint32_t e = var[w]; n += e >= e0 && e < e1; }
        }
        n_in[i] = n;
    }
}
__global__ void __launch_bounds__(256)
k_pidx_range_pairs(const uint32_t* __restrict__ var, const uint32_t* __restrict__ elem_off, uint64_t nb, uint64_t r0, uint64_t var_bytes, uint32_t e0, uint32_t e1,
                   const uint64_t* __restrict__ at_of /* exclusive scan of n_in */, uint64_t pair_base, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * 256) {
        const uint64_t w0 = elem_off[i] >> 2, w1 = (i + 1 < nb ? (uint64_t)elem_off[i + 1] : var_bytes) >> 2;
        // (at_of == nullptr: the range is every edge -- the read's first entry is entry w0 - 2 i of the batch, every element before it
        // having spent two words on offset and lastSkip)
        uint64_t at = pair_base + (at_of ? at_of[i] : w0 - 2 * i);
        for (uint64_t w = w0 + 2; w < w1; ++w) { const uint32_t e = var[w]; if (e >= e0 && e < e1) { keys[at] = e - e0; vals[at] = (uint32_t)(r0 + i); ++at; } }
    }
}

// Stable LSD radix sort of (edge, read) pairs by edge, eight bits a pass.  A tile is RS_ROUNDS x 64 consecutive pairs and
// belongs to ONE wave, which takes it round by round: the lanes with the same digit find each other by eight ballots, rank
// themselves by lane, and a counter per digit in the wave's own LDS carries the rank from round to round -- index order is
// (round, lane), so equal keys keep their order.  Pass 1 counts per tile and digit, a device scan over the digit-major
// counts gives every (digit, tile) its place, pass 2 ranks again and writes.
constexpr int RS_ROUNDS = 32, RS_TILE = 64 * RS_ROUNDS, RS_WAVES = 4;
__device__ __forceinline__ unsigned long long digit_peers(uint32_t d, bool act)
{
    unsigned long long m = __ballot(act);
#pragma unroll
    for (int b = 0; b < 8; ++b) { const unsigned long long v = __ballot(act && ((d >> b) & 1u)); m &= ((d >> b) & 1u) ? v : ~v; }
    return m;
}
__global__ void __launch_bounds__(64 * RS_WAVES)
k_rs_hist(const uint32_t* __restrict__ keys, uint64_t n, uint32_t shift, uint64_t n_tiles, uint64_t* __restrict__ hist /* [256][n_tiles] */)
{
    __shared__ uint32_t cnt[RS_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t tile = (uint64_t)blockIdx.x * RS_WAVES + wave; tile < n_tiles; tile += (uint64_t)gridDim.x * RS_WAVES) {
        for (int i = lane; i < 256; i += 64) cnt[wave][i] = 0;
        wave_sync();
        const uint64_t base = tile * RS_TILE;
        for (int r = 0; r < RS_ROUNDS; ++r) {
            const uint64_t i = base + (uint64_t)r * 64 + lane;
            if (i < n) atomicAdd(&cnt[wave][(keys[i] >> shift) & 255u], 1u);
        }
        wave_sync();
        for (int i = lane; i < 256; i += 64) hist[(uint64_t)i * n_tiles + tile] = cnt[wave][i];
        wave_sync();
    }
}
__global__ void __launch_bounds__(64 * RS_WAVES)
k_rs_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t n, uint32_t shift, uint64_t n_tiles,
             const uint64_t* __restrict__ place /* [256][n_tiles], exclusive scan of hist */, uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out)
{
    __shared__ uint32_t cnt[RS_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t tile = (uint64_t)blockIdx.x * RS_WAVES + wave; tile < n_tiles; tile += (uint64_t)gridDim.x * RS_WAVES) {
        for (int i = lane; i < 256; i += 64) cnt[wave][i] = 0;
        wave_sync();
        const uint64_t base = tile * RS_TILE;
        for (int r = 0; r < RS_ROUNDS; ++r) {
            const uint64_t i = base + (uint64_t)r * 64 + lane;
            const bool act = i < n;
            const uint32_t k = act ? keys[i] : 0u, v = act ? vals[i] : 0u, d = (k >> shift) & 255u;
            const unsigned long long peers = digit_peers(d, act);
            const uint32_t before = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
            uint32_t prior = 0;
            if (act) prior = tld(&cnt[wave][d]);
            wave_sync();                                                   // every lane has read the counters of this round
            if (act && before == 0) tst(&cnt[wave][d], prior + (uint32_t)__popcll(peers));
            wave_sync();
            if (act) { const uint64_t dst = place[(uint64_t)d * n_tiles + tile] + prior + before; keys_out[dst] = k; vals_out[dst] = v; }
        }
        wave_sync();
    }
}

// the file's variable data (unsigned long read ids), element offsets and the per-edge counts of a.countsb
__global__ void __launch_bounds__(256)
k_pidx_widen(const uint32_t* __restrict__ vals, uint64_t n, uint64_t* __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[i] = vals[i];
}
__global__ void __launch_bounds__(256)
k_pidx_tables(const uint32_t* __restrict__ counts, const uint64_t* __restrict__ first /* exclusive scan of counts, [n_he + 1] */, const int32_t* __restrict__ inv,
              uint64_t n_he, uint64_t* __restrict__ elem_off /* [n_he + 1] */, int32_t* __restrict__ countsb)
{
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e <= n_he; e += (uint64_t)gridDim.x * 256) {
        elem_off[e] = 24 + 8 * first[e];
        if (e < n_he) { const int32_t o = inv[e]; countsb[e] = (int32_t)(counts[e] + ((uint64_t)o != e ? counts[o] : 0u)); }
    }
}
__global__ void __launch_bounds__(256)
k_widen_u32(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ out, uint64_t add = 0)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[i] = in[i] + add;
}

// ---- MarkDups (10X/SecretOps.cc:410-566) without the sort: a read's group is its key -- first edge (29 bits), offset on it
// (25 bits, biased), first five bases of its mate (10 bits) -- in an open-address table {key, best}; best = max over the
// group of (quality sum of the pair << 32 | ~read id): the highest sum, the lowest read id among equals -- the member the
// reference keeps after its sort (:529-541).  Every other member marks its pair.  The table covers one `pass` of `n_pass`
// (keys dealt by hash), so that it fits whatever HBM is free.
constexpr unsigned long long DUP_EMPTY = ~0ull;
constexpr int32_t DUP_OFF_BIAS = 1 << 16;
__device__ __forceinline__ uint64_t dup_hash(uint64_t k) { k ^= k >> 31; k *= 0x9E3779B97F4A7C15ull; k ^= k >> 29; k *= 0xBF58476D1CE4E5B9ull; return k ^ (k >> 32); }
// key of read r of the batch, or DUP_EMPTY if it has no path; *score = what atomicMax compares
__device__ __forceinline__ uint64_t dup_key(const uint32_t* __restrict__ var, const uint32_t* __restrict__ elem_off, uint64_t i, uint64_t nb, uint64_t r0,
                                            uint64_t var_bytes, const uint32_t* __restrict__ digest, uint64_t* score, unsigned int* bad, uint64_t id_bias = 0)
{   // (id_bias: a rank of a sharded run numbers its reads from 0; the tie among equal quality sums goes to the lowest id of the WHOLE set)
    const uint64_t w0 = elem_off[i] >> 2, w1 = (i + 1 < nb ? (uint64_t)elem_off[i + 1] : var_bytes) >> 2;
    if (w1 - w0 <= 2) return DUP_EMPTY;
    const int32_t off = (int32_t)var[w0];
    const uint32_t e = var[w0 + 2];
    const uint64_t id = r0 + i, mate = id ^ 1ull;
    if (e >= (1u << 29) || off < -DUP_OFF_BIAS || off >= (1 << 25) - DUP_OFF_BIAS) { atomicOr(bad, 1u); return DUP_EMPTY; }
    const uint32_t dm = digest[mate], ds = digest[id];
    *score = ((uint64_t)((ds >> 10) + (dm >> 10)) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(id + id_bias));
    return ((uint64_t)e << 35) | ((uint64_t)(uint32_t)(off + DUP_OFF_BIAS) << 10) | (uint64_t)(dm & 1023u);
}
template <bool MARK>
__global__ void __launch_bounds__(256)
k_dup_pass(const uint32_t* __restrict__ var, const uint32_t* __restrict__ elem_off, uint64_t nb, uint64_t r0, uint64_t var_bytes,
           const uint32_t* __restrict__ digest, unsigned long long* __restrict__ tkey, unsigned long long* __restrict__ tbest, uint64_t mask,
           uint32_t n_pass, uint32_t pass, uint8_t* __restrict__ dup, unsigned int* __restrict__ bad)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * 256) {
        uint64_t score = 0;
        const uint64_t k = dup_key(var, elem_off, i, nb, r0, var_bytes, digest, &score, bad);
        if (k == DUP_EMPTY) continue;
        const uint64_t h = dup_hash(k);
        if ((uint32_t)(h >> 40) % n_pass != pass) continue;
        uint64_t s = h & mask;
        for (uint64_t guard = 0; guard <= mask; ++guard, s = (s + 1) & mask) {
            if (!MARK) {
                const unsigned long long old = atomicCAS(&tkey[s], DUP_EMPTY, (unsigned long long)k);
                if (old == DUP_EMPTY || old == k) { atomicMax(&tbest[s], (unsigned long long)score); break; }
            } else {
                const unsigned long long cur = tkey[s];
                if (cur == k) { if (tbest[s] != score) dup[(r0 + i) >> 1] = 1; break; }
                if (cur == DUP_EMPTY) { atomicOr(bad, 2u); break; }                       // (every key was inserted by the pass before)
            }
        }
    }
}

} // namespace dfk
