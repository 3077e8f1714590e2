// superplus_amd/csrc/dfk_check_kernels.h -- rows f-1 / f-2 / f-4 checked at sizes no oracle reaches (gfx950, wave64).
//
// Two things, both test infrastructure that lives in the library because the data it looks at stays on the device:
//   * digests of a.paths, a.paths.inv, a.countsb and a.dup that depend on the FILES' content only -- not on the batches the
//     reads were pathed in, the passes the dictionary was counted in, or the entry numbering the k-mer index holds;
//   * k_path_verify: a second look at every placed read that shares no code with path_parts / edit_parts / extend_path
//     (dfk_paths_kernels.h).  It rolls the read's k-mers base by base (KMer::toSuccessor, kmers/KMer.h:189-201, with the
//     reverse complement rolled beside it), finds each in the dictionary with a probe loop of its own, fetches the edge's
//     bases to learn which HBV edge and position that is, and compares with what the read's ReadPath {offset, edges}
//     (paths/long/ReadPath.h:56-63) says about that position.
//     What the reference guarantees and the kernel counts violations of: consecutive edges of a path meet at a vertex
//     (pathPartsToReadPath cuts where they do not, BuildReadQGraph48.cc:1365-1402); the offset lies on the first edge or in
//     front of it; the seed the path was built from is where the path says it is (so every placed read has at least one
//     consistent k-mer).  What it only measures: the share of dictionary hits that agree with the path (a captured gap may
//     shift by up to MAX_JITTER = 3, :657-666; a wrong base can make a k-mer solid somewhere else).
//     tests/path_verify_ref.py computes the same eight counters from the files alone (a.fastb, a.paths, the reads).
#pragma once
#include "dfk_paths_kernels.h"

namespace dfk {

// h over a sequence of values: sum and xor of digest_mix(digest_mix(i + salt) ^ v[i])
template <class T>
__global__ void __launch_bounds__(256)
k_digest_seq(const T* __restrict__ v, uint64_t n, uint64_t salt, unsigned long long* __restrict__ digest /* [0] sum, [1] xor, [2] plain sum of v */)
{
    uint64_t sum = 0, xr = 0, plain = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t x = (uint64_t)v[i];
        const uint64_t h = digest_mix(digest_mix(i + salt) ^ x);
        sum += h; xr ^= digest_mix(h + 0xD1B54A32D192ED03ull); plain += x;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { sum += __shfl_down(sum, d, 64); xr ^= __shfl_down(xr, d, 64); plain += __shfl_down(plain, d, 64); }
    if ((threadIdx.x & 63) == 0 && (sum | xr | plain)) { atomicAdd(&digest[0], sum); atomicXor(&digest[1], xr); atomicAdd(&digest[2], plain); }
}

// a.paths: per read h = chain over (read id, offset, lastSkip, edge ids...) -- the element as the file holds it
__global__ void __launch_bounds__(256)
k_paths_digest(const uint32_t* __restrict__ var, const uint32_t* __restrict__ elem_off, uint64_t nb, uint64_t r0, uint64_t var_bytes,
               unsigned long long* __restrict__ digest)
{
    uint64_t sum = 0, xr = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * 256) {
        const uint64_t w0 = elem_off[i] >> 2, w1 = (i + 1 < nb ? (uint64_t)elem_off[i + 1] : var_bytes) >> 2;
        uint64_t h = digest_mix(r0 + i + 0x9E3779B97F4A7C15ull);
        for (uint64_t w = w0; w < w1; ++w) h = digest_mix(h ^ (uint64_t)var[w]);
        sum += h; xr ^= digest_mix(h + 0xD1B54A32D192ED03ull);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { sum += __shfl_down(sum, d, 64); xr ^= __shfl_down(xr, d, 64); }
    // One slot pair per BLOCK, added to by that block alone (the launches of a run follow each other on one stream): the kernel is
    // launched once per batch -- 187 times at configs[1] -- and a wave's atomics on two shared words, six million of them on the
    // same two addresses, took 0.14 s of a 1.6-s pathing.  k_paths_digest_fold adds the slots up.
    __shared__ uint64_t part[2][4];
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = sum; part[1][threadIdx.x >> 6] = xr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        digest[2 * blockIdx.x] += part[0][0] + part[0][1] + part[0][2] + part[0][3];
        digest[2 * blockIdx.x + 1] ^= part[1][0] ^ part[1][1] ^ part[1][2] ^ part[1][3];
    }
}
__global__ void __launch_bounds__(256)
k_paths_digest_fold(const unsigned long long* __restrict__ slots, uint32_t n_blocks, unsigned long long* __restrict__ out)
{
    uint64_t sum = 0, xr = 0;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += 256) { sum += slots[2 * b]; xr ^= slots[2 * b + 1]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { sum += __shfl_down(sum, d, 64); xr ^= __shfl_down(xr, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], sum); atomicXor(&out[1], xr); }
}

// entries of the paths index that sit on edges equal to their own involution (their reads are counted once in a.countsb)
__global__ void __launch_bounds__(256)
k_self_inverse_sum(const uint32_t* __restrict__ counts, const int32_t* __restrict__ inv, uint64_t n_he, unsigned long long* __restrict__ out)
{
    uint64_t s = 0;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < n_he; e += (uint64_t)gridDim.x * 256) if ((uint64_t)inv[e] == e) s += counts[e];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

__global__ void __launch_bounds__(256)
k_max_u32(const uint32_t* __restrict__ v, uint64_t n, unsigned int* __restrict__ out)
{
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) m = max(m, v[i]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, d, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// the offset tables of a read set handed to dfk_paths_build* (dfk_count's own are checked by k_trim): monotone, inside their
// arrays, and every read's bytes hold its bases
__global__ void __launch_bounds__(256)
k_check_tables(const uint64_t* __restrict__ base_off, uint64_t packed_bytes, const uint32_t* __restrict__ read_len,
               const uint64_t* __restrict__ pq_off, uint64_t pq_bytes, uint64_t n, unsigned int* __restrict__ bad)
{
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (uint64_t)gridDim.x * 256) {
        const uint64_t b0 = base_off[r], b1 = base_off[r + 1];
        if (b1 < b0 || b1 > packed_bytes || b1 - b0 < ((uint64_t)read_len[r] + 3) / 4) atomicOr(bad, 1u);
        if (pq_off) { const uint64_t q0 = pq_off[r], q1 = pq_off[r + 1]; if (q1 < q0 || q1 > pq_bytes) atomicOr(bad, 1u); }
    }
}

enum { PV_PLACED = 0, PV_BROKEN = 1, PV_HITS = 2, PV_CONSISTENT = 3, PV_NO_ANCHOR = 4, PV_ALL_CONSISTENT = 5, PV_DICT_BAD = 6, PV_OUTSIDE = 7, PV_N = 8 };

template <int K>
__global__ void __launch_bounds__(256)
k_path_verify(PartTable pt_arg, const uint32_t* __restrict__ index, uint64_t n_slots, PathGraph G,
              const uint8_t* __restrict__ packed, const uint64_t* __restrict__ base_off, const uint32_t* __restrict__ read_len,
              const uint32_t* __restrict__ var, const uint32_t* __restrict__ elem_off, uint64_t nb, uint64_t r0, uint64_t var_bytes,
              unsigned long long* __restrict__ out /* [PV_N] */)
{
    __shared__ PartLds pt;
    part_lds_init(pt_arg, pt);
    unsigned long long acc[PV_N];
#pragma unroll
    for (int k = 0; k < PV_N; ++k) acc[k] = 0;
    constexpr int BITS = 2 * K, TOP = BITS - 2;
    const u128 m = KTraits<K>::mask();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (uint64_t)gridDim.x * 256) {
        const uint64_t w0 = elem_off[i] >> 2, w1 = (i + 1 < nb ? (uint64_t)elem_off[i + 1] : var_bytes) >> 2;
        const uint32_t cnt = (uint32_t)(w1 - w0) - 2u;
        if (w1 - w0 <= 2) continue;                                        // no path
        ++acc[PV_PLACED];
        const int64_t offset = (int32_t)var[w0];
        const uint32_t* edges = var + w0 + 2;
        // the path itself: ids, consecutive edges meeting at a vertex, the offset on or in front of the first edge
        bool broken = false;
        int64_t total = 0;                                                 // k-mers on the path
        for (uint32_t j = 0; j < cnt && !broken; ++j) {
            const uint32_t e = edges[j];
            if (e >= G.n_he) { broken = true; break; }
            if (j + 1 < cnt) { const uint32_t f = edges[j + 1]; if (f >= G.n_he || G.he_right[e] != G.he_left[f]) broken = true; }
            total += G.ce[G.he_ce[e] >> 1].n;
        }
        if (!broken && offset >= (int64_t)G.ce[G.he_ce[edges[0]] >> 1].n) broken = true;
        if (broken) { ++acc[PV_BROKEN]; continue; }
        // the read's k-mers, rolled base by base, forwards and as reverse complements (both as 2K-bit big-endian numbers)
        const uint64_t r = r0 + i;
        const uint8_t* read = packed + base_off[r];
        const uint32_t n = read_len[r];
        if (n < (uint32_t)K) { ++acc[PV_NO_ANCHOR]; continue; }
        u128 F{0, 0}, R{0, 0};
        uint32_t hits = 0, good = 0;
        for (uint32_t p = 0; p < n; ++p) {
            const uint32_t b = (read[p >> 2] >> (2 * (p & 3))) & 3u;
            F = shl128(F, 2); F.lo |= b; F.lo &= m.lo; F.hi &= m.hi;
            R = shr128(R, 2);
            if (TOP >= 64) R.hi |= (uint64_t)(3u - b) << (TOP - 64); else R.lo |= (uint64_t)(3u - b) << TOP;
            if (p + 1 < (uint32_t)K) continue;
            const uint32_t at = p + 1 - K;                                  // the k-mer's position in the read
            const u128 c = lt128(R, F) ? R : F;
            const u128 kw = shl128(c, 128 - BITS);
            uint64_t s = __umul64hi(set_hash(kw.hi, kw.lo), n_slots);
            uint32_t g = GRAPH_EMPTY;
            uint4 second{0, 0, 0, 0};
            for (uint32_t guard = 0; guard < 1u << 20; ++guard) {
                const uint32_t x = index[s];
                if (x == GRAPH_EMPTY) break;
                const uint4* e = entry_ptr(pt, x);
                const uint4 a = e[0];
                if (((uint64_t)a.x | ((uint64_t)a.y << 32)) == kw.hi && ((uint64_t)a.z | ((uint64_t)a.w << 32)) == kw.lo) { g = x; second = e[1]; break; }
                if (++s == n_slots) s = 0;
            }
            if (g == GRAPH_EMPTY) continue;
            // where the dictionary says this k-mer is: canonical edge, k-mer offset on it as stored
            const uint32_t ce = second.x, off = second.y & 0xFFFFFFu;
            if (ce >= G.n_ce || off >= G.ce[ce].n) { ++acc[PV_DICT_BAD]; continue; }
            const EdgeRec er = G.ce[ce];
            const uint8_t* eb = G.store + er.byte_off;
            bool fw = true, bw = true;                                      // the read's k-mer equals the edge's at `off` / its reverse complement
            for (uint32_t t = 0; t < (uint32_t)K; ++t) {
                const uint32_t x = (eb[(off + t) >> 2] >> (2 * ((off + t) & 3))) & 3u;
                const uint32_t rf = (read[(at + t) >> 2] >> (2 * ((at + t) & 3))) & 3u, rb = 3u - ((read[(at + K - 1 - t) >> 2] >> (2 * ((at + K - 1 - t) & 3))) & 3u);
                fw = fw && x == rf; bw = bw && x == rb;
            }
            if (!fw && !bw) { ++acc[PV_DICT_BAD]; continue; }
            // what the path says about position `at`: path coordinate offset + at, in k-mers along the concatenated edges
            const int64_t coord = offset + (int64_t)at;
            if (coord < 0 || coord >= total) { ++acc[PV_OUTSIDE]; continue; }
            ++hits;
            int64_t before = 0;
            uint32_t j = 0, nk = G.ce[G.he_ce[edges[0]] >> 1].n;
            while (coord >= before + (int64_t)nk) { before += nk; ++j; nk = G.ce[G.he_ce[edges[j]] >> 1].n; }
            const int32_t e_path = (int32_t)edges[j];
            const uint32_t pos = (uint32_t)(coord - before);
            const int2 x = G.xlat[ce];
            const bool ok = (fw && x.x == e_path && off == pos) || (bw && x.y == e_path && er.n - 1u - off == pos);
            good += ok;
        }
        acc[PV_HITS] += hits; acc[PV_CONSISTENT] += good;
        if (!good) ++acc[PV_NO_ANCHOR];
        if (good == hits) ++acc[PV_ALL_CONSISTENT];
    }
#pragma unroll
    for (int k = 0; k < PV_N; ++k) {
        unsigned long long v = acc[k];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&out[k], v);
    }
}

} // namespace dfk
