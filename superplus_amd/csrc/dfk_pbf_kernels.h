// superplus_amd/csrc/dfk_pbf_kernels.h -- SURVEY 8(f)-3: the data-parallel half of ParseBarcodedFastqs on the device (gfx950).
//
// The host keeps what is serial or a library's business -- inflating the .gz files, cutting them into lines, the barcode
// buckets (the iteration order of a std::unordered_set is part of the reference's output, 10X/ParseBarcodedFastqs.cc:311-336).
// The device does, for one group of held pairs:
//   k_pbf_digit + k_rs_*   the order of the pairs: by the place of their barcode in the output, inside a barcode DEScending by
//                          (read 1, read 2) as base sequences, equal pairs in file order -- what the reference's per-barcode
//                          insertion into a list produces (:434-449: a new pair goes in front of the first pair that is not
//                          above it).  A stable LSD radix sort whose digits are read off the sequences themselves: four bases
//                          a digit from the last window of read 2 to the first of read 1 (a sequence that ends reads as A
//                          from there on, and the shorter of two that agree so far is the smaller: its length is the digit
//                          behind its bases), then the barcode's place.  Byte-integer work, HBM-stream bound.
//   k_pbf_pack             bases -> 2-bit codes, LSB-first (N -> A, :407-412; feudal/FieldVec.h:766-770)
//   k_pbf_pq_plan / _emit  PQVecEncoder (feudal/PQVec.cc:18-127): per read, one lane runs the encoder's dynamic programme --
//                          for every prefix the cheapest LAST block, the block list of the longer prefix being the previous
//                          list cut back by what the new block swallows (not a backtrace of the optimum: the procedure is the
//                          result) -- then the blocks are written as the bit stream of :87-127.
#pragma once
#include "dfk_paths_kernels.h"

namespace dfk {

constexpr uint32_t PBF_DROP = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t pbf_code(uint8_t ch, unsigned int* bad)
{
    switch (ch) {
    case 'A': case 'a': case 'N': case 'n': return 0u;
    case 'C': case 'c': return 1u;
    case 'G': case 'g': return 2u;
    case 'T': case 't': return 3u;
    default: atomicOr(bad, 1u); return 0u;
    }
}

struct PbfFiles { const uint8_t* seq[2]; const uint8_t* qual[2]; const uint64_t* off[2]; };

// one digit of the sort key of pair perm[i]: `what` 0 = four bases of read `file` from base 4 * window on (complemented:
// descending), 1 = a byte of its length (complemented), 2 = a byte of the barcode's place (ascending).  Unbarcoded pairs
// (place 0) keep the file's order: all their sequence digits are equal.
__global__ void __launch_bounds__(256)
k_pbf_digit(PbfFiles F, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ perm, uint64_t n, int what, int file, uint32_t window,
            uint32_t* __restrict__ digit, unsigned int* __restrict__ bad)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t p = perm[i], rk = rank[p];
        uint32_t d;
        if (what == 2) d = (rk >> (8 * window)) & 255u;
        else if (rk == 0) d = 0;
        else {
            const uint64_t a = F.off[file][p], L = F.off[file][p + 1] - a;
            if (what == 1) d = 255u - (uint32_t)((L >> (8 * window)) & 255u);
            else {
                uint32_t v = 0;
                for (uint32_t j = 0; j < 4; ++j) { const uint64_t b = 4ull * window + j; v = (v << 2) | (b < L ? pbf_code(F.seq[file][a + b], bad) : 0u); }
                d = 255u - v;
            }
        }
        digit[i] = d;
    }
}

// compaction of the pairs that are written at all, in file order
__global__ void __launch_bounds__(256)
k_pbf_keep(const uint32_t* __restrict__ rank, uint64_t m, uint32_t* __restrict__ perm, unsigned long long* __restrict__ n_out)
{
    // (one block: the order must be the file's; m is a group's pairs, this is a few passes over 4 bytes a pair)
    __shared__ unsigned long long base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (uint64_t i0 = 0; i0 < m; i0 += 256) {
        const uint64_t i = i0 + threadIdx.x;
        const bool keep = i < m && rank[i] != PBF_DROP;
        const unsigned long long mk = __ballot(keep);
        __shared__ uint32_t wcount[4];
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) wcount[wave] = (uint32_t)__popcll(mk);
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wcount[w];
        if (keep) perm[base + before + __popcll(mk & ((1ull << lane) - 1ull))] = (uint32_t)i;
        __syncthreads();
        if (threadIdx.x == 0) base += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = base;
}

// output read k = read (k & 1) of pair order[k >> 1]: its length, its bytes in .fastb
__global__ void __launch_bounds__(256)
k_pbf_sizes(PbfFiles F, const uint32_t* __restrict__ order, uint64_t n_reads, uint32_t* __restrict__ len, uint64_t* __restrict__ fastb_sz /* [n_reads + 1] */)
{
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k <= n_reads; k += (uint64_t)gridDim.x * 256) {
        if (k == n_reads) { fastb_sz[k] = 0; continue; }
        const uint32_t p = order[k >> 1]; const int f = (int)(k & 1);
        const uint64_t L = F.off[f][p + 1] - F.off[f][p];
        len[k] = (uint32_t)L; fastb_sz[k] = (L + 3) >> 2;
    }
}

__global__ void __launch_bounds__(256)
k_pbf_pack(PbfFiles F, const uint32_t* __restrict__ order, uint64_t n_reads, const uint64_t* __restrict__ fastb_off, uint8_t* __restrict__ var, unsigned int* __restrict__ bad)
{
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n_reads; k += (uint64_t)gridDim.x * 256) {
        const uint32_t p = order[k >> 1]; const int f = (int)(k & 1);
        const uint64_t a = F.off[f][p], L = F.off[f][p + 1] - a;
        uint8_t* o = var + fastb_off[k];
        for (uint64_t j = 0; j < L; j += 4) {
            uint32_t v = 0;
            for (uint32_t t = 0; t < 4 && j + t < L; ++t) v |= pbf_code(F.seq[f][a + j + t], bad) << (2 * t);
            o[j >> 2] = (uint8_t)v;
        }
    }
}

// ---- PQVecEncoder
struct PbfBlock { uint8_t n, bits, minq, pad; };
__device__ __forceinline__ uint32_t pbf_ceil_lg2(uint32_t v) { return v <= 1u ? 0u : 32u - (uint32_t)__clz(v - 1u); }
__device__ __forceinline__ uint32_t pbf_block_size(uint32_t n, uint32_t bits) { return (n * bits + 17u + 7u) >> 3; }

// scratch of read k: cost[L + 1] then blocks[L] (4 bytes each), at scratch_off[k] words
__global__ void __launch_bounds__(128)
k_pbf_pq_plan(PbfFiles F, const uint32_t* __restrict__ order, uint64_t k0, uint64_t nk, const uint64_t* __restrict__ scratch_off, uint32_t* __restrict__ scratch,
              uint32_t* __restrict__ n_blocks, uint64_t* __restrict__ pq_sz /* this batch: [nk] */, unsigned int* __restrict__ bad)
{
    for (uint64_t t = (uint64_t)blockIdx.x * 128 + threadIdx.x; t < nk; t += (uint64_t)gridDim.x * 128) {
        const uint64_t k = k0 + t;
        const uint32_t p = order[k >> 1]; const int f = (int)(k & 1);
        const uint64_t a = F.off[f][p];
        const uint32_t L = (uint32_t)(F.off[f][p + 1] - a);
        const uint8_t* q = F.qual[f] + a;
        uint32_t* cost = scratch + scratch_off[t];
        PbfBlock* blocks = reinterpret_cast<PbfBlock*>(cost + L + 1);
        cost[0] = 1;
        uint32_t nb = 0;
        for (uint32_t i = 0; i < L; ++i) {
            const uint32_t qi = (uint32_t)q[i] - 33u;
            if (qi > 63u) atomicOr(bad, 2u);
            uint32_t mn = qi, mx = qi, bits = 0, n = 1;
            uint32_t best_cost = cost[i] + pbf_block_size(1, 0);
            PbfBlock best{1, 0, (uint8_t)mn, 0};
            for (uint32_t j = i; j > 0 && n < 255u;) {
                const uint32_t v = ((uint32_t)q[--j] - 33u) & 255u;
                mx = max(mx, v); mn = min(mn, v);
                bits = pbf_ceil_lg2(mx + 1u - mn);
                const uint32_t c = cost[j] + pbf_block_size(++n, bits);
                if (c < best_cost) { best_cost = c; best = PbfBlock{(uint8_t)n, (uint8_t)bits, (uint8_t)mn, 0}; }
            }
            cost[i + 1] = best_cost;
            uint32_t remove = best.n - 1u;
            if (!remove) blocks[nb++] = best;
            else {
                while (remove > blocks[nb - 1].n) { remove -= blocks[nb - 1].n; --nb; }
                if (remove == blocks[nb - 1].n) blocks[nb - 1] = best;
                else { blocks[nb - 1].n = (uint8_t)(blocks[nb - 1].n - remove); blocks[nb++] = best; }
            }
        }
        uint64_t bytes = 1;                                              // the terminating 0
        for (uint32_t b = 0; b < nb; ++b) bytes += pbf_block_size(blocks[b].n, blocks[b].bits);
        n_blocks[t] = nb; pq_sz[t] = bytes;
    }
}

// The same plan for reads of at most PBF_LDS_LEN values, one wave per workgroup: the encoder's two arrays -- the best cost of
// every prefix and the qualities -- live in LDS, TRANSPOSED ([position][lane]: the 64 reads of a wave walk j down from i together,
// so a wave's read of cost[j] is 64 consecutive halfwords, conflict-free), where the version above keeps them in each read's own
// scratch in HBM: 5000 dependent, uncoalesced loads per read of 100 values, 713 ms for 4 M reads against 40 here.  The block
// list (one push or cut-back per position) stays in the read's scratch.  A prefix of <= 256 values costs < 1024 bytes: 16 bits.
constexpr uint32_t PBF_LDS_LEN = 256;
__global__ void __launch_bounds__(64)
k_pbf_pq_plan_lds(PbfFiles F, const uint32_t* __restrict__ order, uint64_t k0, uint64_t nk, const uint64_t* __restrict__ scratch_off, uint32_t* __restrict__ scratch,
                  uint32_t* __restrict__ n_blocks, uint64_t* __restrict__ pq_sz, unsigned int* __restrict__ bad)
{
    __shared__ uint16_t cost[PBF_LDS_LEN + 1][64];
    __shared__ uint8_t qq[PBF_LDS_LEN][64];
    const int lane = threadIdx.x;
    for (uint64_t t0 = (uint64_t)blockIdx.x * 64; t0 < nk; t0 += (uint64_t)gridDim.x * 64) {
        const uint64_t t = t0 + lane;
        const bool live = t < nk;
        const uint64_t k = k0 + (live ? t : nk - 1);
        const uint32_t p = order[k >> 1]; const int f = (int)(k & 1);
        const uint64_t a = F.off[f][p];
        const uint32_t L = live ? (uint32_t)(F.off[f][p + 1] - a) : 0u;
        const uint8_t* q = F.qual[f] + a;
        PbfBlock* blocks = reinterpret_cast<PbfBlock*>(scratch + scratch_off[live ? t : nk - 1] + L + 1);
        for (uint32_t j = 0; j < L; ++j) {
            const uint32_t v = (uint32_t)q[j] - 33u;
            if (v > 63u) atomicOr(bad, 2u);
            qq[j][lane] = (uint8_t)v;
        }
        cost[0][lane] = 1;
        uint32_t Lmax = L;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) Lmax = max(Lmax, (uint32_t)__shfl_xor((int)Lmax, d, 64));
        uint32_t nb = 0;
        PbfBlock top{0, 0, 0, 0};                                          // blocks[nb - 1], kept in registers
        for (uint32_t i = 0; i < Lmax; ++i) {
            if (i >= L) continue;
            const uint32_t qi = qq[i][lane];
            uint32_t mn = qi, mx = qi, n = 1;
            uint32_t best_cost = (uint32_t)cost[i][lane] + pbf_block_size(1, 0);
            uint32_t best = 1u | (0u << 8) | (mn << 16);                   // n | bits << 8 | minq << 16
            for (uint32_t j = i; j > 0 && n < 255u;) {
                --j;
                const uint32_t v = qq[j][lane];
                mx = max(mx, v); mn = min(mn, v);
                const uint32_t bits = pbf_ceil_lg2(mx + 1u - mn);
                const uint32_t c = (uint32_t)cost[j][lane] + pbf_block_size(++n, bits);
                if (c < best_cost) { best_cost = c; best = n | (bits << 8) | (mn << 16); }
            }
            cost[i + 1][lane] = (uint16_t)best_cost;
            const PbfBlock B{(uint8_t)best, (uint8_t)(best >> 8), (uint8_t)(best >> 16), 0};
            uint32_t remove = B.n - 1u;
            if (!remove) { if (nb) blocks[nb - 1] = top; top = B; ++nb; }
            else {
                while (remove > top.n) { remove -= top.n; --nb; top = blocks[nb - 1]; }
                if (remove == top.n) top = B;
                else { top.n = (uint8_t)(top.n - remove); blocks[nb - 1] = top; top = B; ++nb; }
            }
        }
        if (live) {
            if (nb) blocks[nb - 1] = top;
            uint64_t bytes = 1;
            for (uint32_t b = 0; b < nb; ++b) bytes += pbf_block_size(blocks[b].n, blocks[b].bits);
            n_blocks[t] = nb; pq_sz[t] = bytes;
        }
    }
}

__global__ void __launch_bounds__(128)
k_pbf_pq_emit(PbfFiles F, const uint32_t* __restrict__ order, uint64_t k0, uint64_t nk, const uint64_t* __restrict__ scratch_off, const uint32_t* __restrict__ scratch,
              const uint32_t* __restrict__ n_blocks, const uint64_t* __restrict__ pq_off /* global: [n_reads + 1] */, uint8_t* __restrict__ var)
{
    for (uint64_t t = (uint64_t)blockIdx.x * 128 + threadIdx.x; t < nk; t += (uint64_t)gridDim.x * 128) {
        const uint64_t k = k0 + t;
        const uint32_t p = order[k >> 1]; const int f = (int)(k & 1);
        const uint64_t a = F.off[f][p];
        const uint32_t L = (uint32_t)(F.off[f][p + 1] - a);
        const uint8_t* it = F.qual[f] + a;
        const PbfBlock* blocks = reinterpret_cast<const PbfBlock*>(scratch + scratch_off[t] + L + 1);
        uint8_t* o = var + pq_off[k];
        for (uint32_t b = 0; b < n_blocks[t]; ++b) {
            const PbfBlock B = blocks[b];
            *o++ = B.n;
            uint64_t acc = (uint64_t)B.bits | ((uint64_t)B.minq << 3);
            *o++ = (uint8_t)acc; acc >>= 8;
            if (!B.bits) { *o++ = (uint8_t)acc; it += B.n; continue; }
            uint32_t off = 1;
            for (uint32_t x = 0; x < B.n; ++x) {
                acc |= (uint64_t)(((uint32_t)*it++ - 33u) - B.minq) << off;
                if ((off += B.bits) >= 8u) { *o++ = (uint8_t)acc; off -= 8u; acc >>= 8; }
            }
            if (off) *o++ = (uint8_t)acc;
        }
        *o = 0;
    }
}

__global__ void __launch_bounds__(256)
k_pbf_scratch_sizes(PbfFiles F, const uint32_t* __restrict__ order, uint64_t k0, uint64_t nk, uint64_t* __restrict__ sz /* [nk + 1] */)
{
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t <= nk; t += (uint64_t)gridDim.x * 256) {
        if (t == nk) { sz[t] = 0; continue; }
        const uint64_t k = k0 + t;
        const uint32_t p = order[k >> 1]; const int f = (int)(k & 1);
        const uint64_t L = F.off[f][p + 1] - F.off[f][p];
        sz[t] = 2 * L + 1;                                                // cost[L + 1] + blocks[L], in 4-byte words
    }
}

__global__ void __launch_bounds__(256)
k_pbf_iota(uint32_t* __restrict__ p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) p[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256)
k_pbf_max_len(const uint64_t* __restrict__ off, uint64_t m, unsigned long long* __restrict__ out)
{
    unsigned long long mx = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (uint64_t)gridDim.x * 256) mx = max(mx, (unsigned long long)(off[i + 1] - off[i]));
    atomicMax(out, mx);
}

} // namespace dfk
