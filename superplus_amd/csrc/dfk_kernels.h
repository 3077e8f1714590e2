// superplus_amd/csrc/dfk_kernels.h -- the dfk HIP kernels (gfx950 / CDNA4, wave64).
// Integer/hash work only: no MFMA anywhere.  See DESIGN.md for the data layout in HBM and
// the roofline of each kernel.
#pragma once
#include <type_traits>
#include "dfk_device.h"

namespace dfk {

// ============================================================================ a1: quality-tail trim
// One lane per read.  Streams the PQVec blocks forward (feudal/PQVec.cc:87-127 layout) and
// keeps the end of the right-most run of >= K quals >= min_qual, which is what
// GoodLenTailFinder finds scanning from the 3' end (BuildReadQGraph48.cc:70-80).
// Also accumulates the instance total  sum max(0, goodLen-K+1 | goodLen >= K+1)  and
// flags reads whose PQVec length disagrees with read_len.
template <int K>
__global__ void __launch_bounds__(256)
k_trim(const uint8_t* __restrict__ pq, const uint64_t* __restrict__ pq_off, uint64_t pq_nbytes,
       const uint64_t* __restrict__ base_off, uint64_t packed_bytes, const uint32_t* __restrict__ read_len,
       uint64_t n_reads, uint32_t min_qual, uint32_t* __restrict__ good_len,
       unsigned long long* __restrict__ n_inst, unsigned int* __restrict__ bad,
       unsigned long long* __restrict__ sum_good)
{
    // grid-stride: the totals are reduced per block, so the three global atomics are issued once per
    // block (same-address atomics run at ~88 per microsecond on this chip; one per wave of reads cost 10 ms)
    unsigned long long mine = 0, good = 0;
    bool any_bad = false;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t gl = 0;
        uint64_t p = pq_off[r], end = pq_off[r + 1];
        uint32_t idx = 0, run = 0;
        // the offset tables are the caller's: nothing is read through them before they have been checked against
        // the array sizes (a malformed .qualp/.fastb must end as DFK_E_INPUT, not as a stray device read)
        const uint64_t b0 = base_off[r], b1 = base_off[r + 1];
        bool ok = p <= end && end <= pq_nbytes && b0 <= b1 && b1 <= packed_bytes && b1 - b0 >= ((uint64_t)read_len[r] + 3) / 4;
        if (!ok) end = p;
        while (p < end) {
            uint32_t nQs = pq[p];
            if (!nQs) break;
            if (p + 3 > end) { ok = false; break; }
            uint32_t hdr = pq[p + 1] | ((uint32_t)pq[p + 2] << 8);
            uint32_t nBits = hdr & 7u, minQ = (hdr >> 3) & 63u;
            uint64_t blk = ((uint64_t)nQs * nBits + 24) >> 3;
            if (p + blk > end) { ok = false; break; }
            if (nBits == 0) {
                if (minQ >= min_qual) { run += nQs; if (run >= (uint32_t)K) gl = idx + nQs; }
                else run = 0;
                idx += nQs;
            } else {
                // the block's quality fields, 64 bits at a time (two byte loads per base, as it was first written, made
                // a read with per-base quality blocks ten times as expensive as one with a single run: 165 ms of trim
                // for a quarter of 1.8 G reads)
                uint64_t bit = 8 * (p + 1) + 9;
                const uint32_t fmask = (1u << nBits) - 1u;
                for (uint32_t left = nQs; left;) {
                    const uint64_t by = bit >> 3;
                    const uint32_t sh = (uint32_t)(bit & 7);
                    uint64_t w = 0;
                    if (by + 8 <= pq_nbytes) __builtin_memcpy(&w, pq + by, 8);
                    else for (uint32_t k = 0; k < 8 && by + k < pq_nbytes; ++k) w |= (uint64_t)pq[by + k] << (8 * k);
                    w >>= sh;
                    const uint32_t n = min(left, (64u - sh) / nBits);
                    for (uint32_t i = 0; i < n; ++i, w >>= nBits) {
                        const uint32_t q = minQ + ((uint32_t)w & fmask);
                        if (q < min_qual) run = 0;
                        else if (++run >= (uint32_t)K) gl = idx + 1;
                        ++idx;
                    }
                    bit += (uint64_t)n * nBits; left -= n;
                }
            }
            p += blk;
        }
        uint32_t rl = read_len[r];
        if (!ok || idx != rl) { any_bad = true; gl = 0; }
        if (gl > rl) gl = rl;
        good_len[r] = gl;
        good += gl;
        if (gl >= (uint32_t)K + 1) mine += gl - K + 1;
    }
    __shared__ unsigned long long sh[2][4];
    __shared__ unsigned int sh_bad;
    if (threadIdx.x == 0) sh_bad = 0;
    __syncthreads();
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { mine += __shfl_down(mine, d, 64); good += __shfl_down(good, d, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = mine; sh[1][threadIdx.x >> 6] = good; }
    if (any_bad) atomicOr(&sh_bad, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a0 = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3], a1 = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        if (a0) atomicAdd(n_inst, a0);
        if (a1) atomicAdd(sum_good, a1);
        if (sh_bad) atomicOr(bad, 1u);
    }
}

// ============================================================================ a2 (first half): super-k-mer partition
// One lane per read.  For every k-mer position s the fine bucket is a hash of the minimum
// canonical-m-mer hash over the k-mer's W = K-M+1 m-mers (a property of the canonical k-mer:
// a k-mer and its reverse complement hold the same canonical m-mers).  The sliding minimum
// uses the block prefix/suffix scheme so every lane does the same work per base (no
// data-dependent rescans, which a wave would pay for on every step): m-mer positions are cut
// into blocks of W; `arr` (LDS, W words per lane, lane-minor so it is conflict-free) holds
// the suffix minima of the previous block and is overwritten in place with the hashes of the
// current one.
//   WRITE == false: add (1<<32 | nk) to bucket_acc[bucket] per record (records and instances
//                   per fine bucket).
//   WRITE == true : records are queued per lane (LDS) and written after the scan: 32 bytes =
//                   {header, barcode, 192 bits of the read's own 2-bit stream starting one base
//                   before the run}.
struct PartParams {
    uint32_t M;            // minimizer length (<= 16)
    uint32_t W;            // K - M + 1
    uint32_t log2_nb;      // fine buckets = 1 << log2_nb
    uint32_t log2_world;   // fine buckets are laid out owner-major: owner rank = bucket & (world-1)
    int64_t  read_id0;     // global index of this shard's first read (ignBcBelow compares global read ids)
    uint32_t sub_lo;       // pass: fine buckets whose rank-local id (global id without the owner bits) is in
    uint32_t sub_n;        // [sub_lo, sub_lo + sub_n); numbered owner * sub_n + (id - sub_lo) inside the pass
    uint32_t class_mask;   // the SWEEP_CLASSES-ths of an owner's bucket space that [sub_lo, sub_lo + sub_n) touches (bit per class)
};

// The counting scan leaves, per read, a bit mask of the coarse classes its runs' buckets fall in (32 equal
// ranges of an owner's bucket space).  A scatter sweep tests it against the classes its pass touches and does
// not look at the other reads at all: a read has ~4 runs, so a pass over 1/8 of the buckets needs ~45 % of the
// reads, one over 1/32 of them 11 %.
constexpr uint32_t SWEEP_CLASSES = 32;
// -DDFK_RUN_CLASSES (measured, not the default): behind the masks, in the same buffer, per read one 64-bit word with the
// class of EACH of its (up to ten) runs, five bits a run.  A sweep then rebuilds the bucket -- a random read of the
// minimizer's bases and a hash -- only for the runs whose class the pass touches instead of for every run of every listed
// read (that look-up is four times as frequent as the records a sweep writes).  At configs[1]: a sweep takes 81.7 ms
// instead of 95.9 -- but the 8 bytes per read (14.4 GB) make it 11 passes instead of 9, and the step comes out the same
// (1504 against 1508 ms); the sweeps hide under the counts either way.
#ifndef DFK_RUN_CLASSES
__host__ __device__ inline uint64_t read_classes_bytes(uint64_t n_reads) { return n_reads * 4; }
__device__ __forceinline__ const uint64_t* run_classes_of(const uint32_t*, uint64_t) { return nullptr; }
#else
__host__ __device__ inline uint64_t read_classes_bytes(uint64_t n_reads) { return ((n_reads + 1) & ~1ull) * 4 + n_reads * 8; }
__device__ __forceinline__ const uint64_t* run_classes_of(const uint32_t* read_classes, uint64_t n_reads)
{ return read_classes ? reinterpret_cast<const uint64_t*>(read_classes + ((n_reads + 1) & ~1ull)) : nullptr; }
#endif
__host__ __device__ inline uint32_t sweep_class_of(uint32_t local_bucket, uint32_t log2_local)
{ return log2_local > 5 ? local_bucket >> (log2_local - 5) : local_bucket; }
__host__ __device__ inline uint32_t sweep_class_mask(uint32_t sub_lo, uint32_t sub_n, uint32_t log2_local)
{
    if (sub_n == 0) return 0u;
    const uint32_t a = sweep_class_of(sub_lo, log2_local), b = sweep_class_of(sub_lo + sub_n - 1, log2_local);
    const uint32_t upto_b = b >= 31 ? 0xFFFFFFFFu : ((1u << (b + 1)) - 1u);
    return upto_b & ~((1u << a) - 1u);
}

constexpr int PART_THREADS = 128;
constexpr int PART_RING = 2;          // words of its read a lane of the counting scan keeps staged in LDS
// Per-read run summary written by the counting scan (16 bytes): bits 0-3 = number of runs (super-k-mers) or
// SUMMARY_OVERFLOW, then from bit 8 twelve bits per run: nk (6) | offset of its minimizer from the run's
// first k-mer (6, < W).  The scatter passes rebuild each run's bucket from the 2M bits at that offset
// instead of scanning the read again.
constexpr int SUMMARY_RUNS = 10;
constexpr uint32_t SUMMARY_OVERFLOW = 15;

// pass-local number of a global fine bucket id, or ~0 if the bucket is not in this pass
__device__ __forceinline__ uint32_t pass_local(uint32_t bucket, const PartParams& pp)
{
    const uint32_t ls = pp.log2_nb - pp.log2_world;
    const uint32_t d = (bucket & ((1u << ls) - 1u)) - pp.sub_lo;
    return d < pp.sub_n ? (bucket >> ls) * pp.sub_n + d : 0xFFFFFFFFu;
}

// fine bucket of a minimizer hash: global id, owner-major (all fine buckets of one owner rank contiguous)
__device__ __forceinline__ uint32_t bucket_of(uint32_t minhash, const PartParams& pp)
{
    const uint32_t b = (minhash * 0x9E3779B1u) >> (32 - pp.log2_nb);
    return ((b & ((1u << pp.log2_world) - 1u)) << (pp.log2_nb - pp.log2_world)) | (b >> pp.log2_world);
}
constexpr int PART_QCAP = 8;          // queued records per lane before an early flush

template <int K>
__device__ __forceinline__ void emit_record(const uint32_t* __restrict__ words, uint64_t n_words, uint64_t bit0,
                                            uint32_t s0, uint32_t nk, uint32_t bucket, uint32_t gl, int32_t tag,
                                            uint64_t dst_index, uint4* __restrict__ records)
{
    const bool hp = s0 > 0;
    const bool hs = s0 + nk + (uint32_t)K <= gl;                 // (callers that do not know gl pass 0xFFFFFFFF or 0: succ / no succ)
    // bases [s0-1, s0+nk+K) of the read (slot 0 is a dummy when the run starts the read)
    uint64_t bo = bit0 + 2ull * (hp ? s0 - 1 : 0);
    uint64_t wi = bo >> 5; uint32_t sh = (uint32_t)bo & 31u;
    uint32_t w[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) w[i] = (wi + i < n_words) ? words[wi + i] : 0u;
    uint32_t o[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = alignbit(w[i + 1], w[i], sh);
    if (!hp) {   // shift the stream up one base; dummy predecessor = 0
#pragma unroll
        for (int i = 5; i > 0; --i) o[i] = (o[i] << 2) | (o[i - 1] >> 30);
        o[0] <<= 2;
    }
    uint32_t nbits = 2u * (1u + nk + (uint32_t)K - 1u + (hs ? 1u : 0u));   // <= 192
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        uint32_t lo = 32u * i;
        if (nbits <= lo) o[i] = 0;
        else if (nbits < lo + 32u) o[i] &= (1u << (nbits - lo)) - 1u;
    }
    uint4 a{rec_header(nk, hp, hs, bucket), (uint32_t)tag, o[0], o[1]};
    uint4 b{o[2], o[3], o[4], o[5]};
    records[2 * dst_index] = a;
    records[2 * dst_index + 1] = b;
}

// Sharded runs do not need per-bucket counts on the sending side (the owner regroups what it receives), only
// how many records go to each owner in each pass: bucket space of an owner cut into PART_CLASSES equal
// classes; a pass is a whole number of classes.
constexpr uint32_t PART_CLASSES = 64;
__device__ __forceinline__ uint32_t class_bin(uint32_t bucket, const PartParams& pp)
{
    const uint32_t ls = pp.log2_nb - pp.log2_world;
    const uint32_t sub = bucket & ((1u << ls) - 1u);
    return (bucket >> ls) * PART_CLASSES + (ls > 6 ? sub >> (ls - 6) : sub);
}

// the scan of one read (lane per read; no block barriers inside)
template <int K, bool WRITE>
__device__ __forceinline__ void
partition_read(uint64_t r, uint32_t* smem, uint32_t* __restrict__ lh,
            const uint8_t* __restrict__ packed, uint64_t packed_bytes, const uint64_t* __restrict__ base_off,
            const uint32_t* __restrict__ good_len, const int32_t* __restrict__ bc, int64_t ign_bc_below,
            uint64_t n_reads, const PartParams& pp,
            unsigned long long* __restrict__ bucket_acc, unsigned long long* __restrict__ bucket_cur, uint64_t n_out,
            uint4* __restrict__ records, uint4* __restrict__ summaries, const uint32_t* __restrict__ read_list, uint64_t n_list,
            const uint64_t* __restrict__ slice_base, uint32_t* __restrict__ read_classes)
{
    // where a record of pass-local bucket b goes: its bucket's cursor, or (sharded) its owner's slice
    auto place = [&](uint32_t b) -> uint64_t {
        if (!slice_base) { const uint64_t d = atomicAdd(&bucket_cur[b], 1ull); return d < n_out ? d : ~0ull; }
        const uint32_t owner = b / pp.sub_n;
        const uint64_t d = slice_base[owner] + atomicAdd(&bucket_cur[owner], 1ull);
        return d < slice_base[owner + 1] ? d : ~0ull;
    };
    uint32_t* arr = smem;                                        // [W][PART_THREADS] hashes -> suffix minima
    uint32_t* queue = smem + pp.W * PART_THREADS;                // WRITE: [PART_QCAP][2][PART_THREADS]
    uint8_t* sidx = reinterpret_cast<uint8_t*>(smem + pp.W * PART_THREADS);              // !WRITE: [W][PART_THREADS] where each suffix minimum sits
    uint32_t* ring = reinterpret_cast<uint32_t*>(sidx + pp.W * PART_THREADS);            // !WRITE: [PART_RING][PART_THREADS] words of the read
    uint64_t sum_lo = 0, sum_hi = 0;                                                     // !WRITE: the run fields of the summary, packed as they come
    const int tid = threadIdx.x;
    if (WRITE && read_list) { if (r >= n_list) return; r = read_list[r]; }
    const uint32_t M = pp.M, W = pp.W;
    const uint32_t gl = r < n_reads ? good_len[r] : 0;
    if (gl < (uint32_t)K + 1) {                                  // Kmerizer::map: len < K+1 emits nothing (:153)
        if (!WRITE && r < n_reads) { summaries[r] = uint4{0, 0, 0, 0}; if (read_classes) { read_classes[r] = 0u; if (uint64_t* rc_ = const_cast<uint64_t*>(run_classes_of(read_classes, n_reads))) rc_[r] = 0ull; } }
        return;
    }
    uint32_t Pi = 0, cur_rel = 0;                                // !WRITE: where the prefix minimum sits; minimizer offset of the open run
    uint32_t cmask = 0;                                          // !WRITE: classes of the runs' buckets (the sweeps' prefilter)
    uint64_t rcls = 0;                                           // !WRITE: ... and run by run, five bits each
    const uint32_t log2_local = pp.log2_nb - pp.log2_world;

    const uint64_t byte0 = base_off[r];
    const uint32_t* words = reinterpret_cast<const uint32_t*>(packed);   // hipMalloc'd: 4-byte aligned base
    const uint64_t n_words = (packed_bytes + 3) >> 2;
    const uint64_t bit0 = byte0 * 8;                             // bit offset of base 0 in the word stream
    int32_t tag = -1;                                            // :150-151
    if (bc && (int64_t)r + pp.read_id0 >= ign_bc_below) tag = bc[r];

    const uint32_t mmask = M == 16 ? 0xFFFFFFFFu : ((1u << (2 * M)) - 1u);
    const uint32_t rsh = 2 * (M - 1);
    uint32_t f = 0, rc = 0;                 // forward / reverse-complement m-mer, big-endian
    uint32_t P = 0;                          // prefix minimum inside the current block
    uint32_t bi = 0;                         // index inside the current block of m-mer positions
    uint32_t cur_b = 0, cur_s0 = 0, cur_nk = 0;
    uint32_t qn = 0;

    // the word after the current one is always in flight (a load issued when it is needed costs a full
    // memory latency every 16 bases)
    // The counting scan stages PART_RING words of the read in LDS at a time (one fill covers a 100-base read): a
    // global load inside the loop has to be waited for with vmcnt(0), which also waits for every bucket atomic
    // the lane has in flight -- that coupling cost the scan 200 ms of 560.  The scanning scatter (rare) keeps
    // the word after the current one in a register instead.
    uint64_t wi = bit0 >> 5;
    uint64_t filled_to = wi;
    auto fill = [&]() {
        uint32_t w[PART_RING];
#pragma unroll
        for (int k = 0; k < PART_RING; ++k) w[k] = filled_to + k < n_words ? words[filled_to + k] : 0u;
#pragma unroll
        for (int k = 0; k < PART_RING; ++k) ring[((filled_to + k) & (PART_RING - 1)) * PART_THREADS + tid] = w[k];
        filled_to += PART_RING;
    };
    uint32_t wbits, wnext = 0;
    if (!WRITE) { fill(); wbits = ring[(wi & (PART_RING - 1)) * PART_THREADS + tid]; }
    else { wbits = wi < n_words ? words[wi] : 0u; wnext = wi + 1 < n_words ? words[wi + 1] : 0u; }
    uint32_t wpos = (uint32_t)bit0 & 31u;
    uint32_t sfx_next = 0xFFFFFFFFu;        // arr[bi + 1] of the previous block, read one position ahead

    auto close_run = [&]() {
        if (WRITE && cur_b == 0xFFFFFFFFu) return;                  // run belongs to another pass
        if (!WRITE) {
            if (lh) { const uint32_t bin = class_bin(cur_b, pp); atomicAdd(&lh[bin], 1u); atomicAdd(&lh[(PART_CLASSES << pp.log2_world) + bin], cur_nk); }
            else atomicAdd(&bucket_acc[cur_b], (1ull << 32) | cur_nk);
            const uint32_t cls = sweep_class_of(cur_b & ((1u << log2_local) - 1u), log2_local);
            cmask |= 1u << cls;
            if (qn < (uint32_t)SUMMARY_RUNS) rcls |= (uint64_t)cls << (5u * qn);
            if (qn < (uint32_t)SUMMARY_RUNS) {                          // 12 bits per run from bit 8 (kept in registers: LDS is what limits this kernel's occupancy)
                const uint64_t fld = cur_nk | (cur_rel << 6);
                const uint32_t b = 8u + 12u * qn;
                if (b < 64u) { sum_lo |= fld << b; if (b > 52u) sum_hi |= fld >> (64u - b); }
                else sum_hi |= fld << (b - 64u);
            }
            ++qn;
        } else {
            if (qn == PART_QCAP) {        // rare: flush early
                for (uint32_t e = 0; e < qn; ++e) {
                    uint32_t a = queue[(2 * e) * PART_THREADS + tid], b = queue[(2 * e + 1) * PART_THREADS + tid];
                    const uint64_t dst = place(b);
                    if (dst != ~0ull) emit_record<K>(words, n_words, bit0, a & 0xFFFFu, a >> 16, b, gl, tag, dst, records);
                }
                qn = 0;
            }
            queue[(2 * qn) * PART_THREADS + tid] = cur_s0 | (cur_nk << 16);
            queue[(2 * qn + 1) * PART_THREADS + tid] = cur_b;
            ++qn;
        }
    };

    for (uint32_t j = 0; j < gl; ++j) {
        uint32_t b = (wbits >> wpos) & 3u;
        wpos += 2;
        if (wpos == 32) {
            wpos = 0; ++wi;
            if (!WRITE) { if (wi == filled_to) fill(); wbits = ring[(wi & (PART_RING - 1)) * PART_THREADS + tid]; }
            else { wbits = wnext; wnext = wi + 1 < n_words ? words[wi + 1] : 0u; }
        }
        f = ((f << 2) | b) & mmask;
        rc = (rc >> 2) | ((3u - b) << rsh);
        if (j + 1 < M) continue;
        uint32_t h = mmer_hash(f < rc ? f : rc);
        const bool newmin = bi == 0 || h < P;
        P = newmin ? h : P;
        if (!WRITE) Pi = newmin ? bi : Pi;
        const uint32_t t = j + 1 - M;                           // m-mer position
        if (t + 1 >= W) {                                        // k-mer s = t-W+1 is complete
            uint32_t mv = P;
            bool older = false;                                  // the minimum sits in the previous block
            if (bi != W - 1) { const uint32_t sfx = sfx_next; older = sfx < mv; mv = older ? sfx : mv; }
            const uint32_t s = t + 1 - W;
            uint32_t bucket = bucket_of(mv, pp);
            if (WRITE) bucket = pass_local(bucket, pp);           // buckets of other passes become "no bucket"
            const bool open = s == 0 || bucket != cur_b || cur_nk == (uint32_t)KTraits<K>::NK_MAX;
            if (open) {
                if (s != 0) close_run();
                cur_b = bucket; cur_s0 = s; cur_nk = 1;
                // offset (from s) of an m-mer whose hash is the window minimum: block index i of the previous
                // block is position s + i - bi - 1, index i of the current one is position s + W - 1 - bi + i
                if (!WRITE) cur_rel = older ? (uint32_t)sidx[(bi + 1) * PART_THREADS + tid] - bi - 1u : W - 1u - bi + Pi;
            } else ++cur_nk;
        }
        arr[bi * PART_THREADS + tid] = h;
        if (++bi == W) {                                         // block complete: turn it into suffix minima
            uint32_t run = h, ri = W - 1;
            if (!WRITE) sidx[(W - 1) * PART_THREADS + tid] = (uint8_t)(W - 1);
            // eight reads in flight at a time: read-compare-write one element after the other is a chain of
            // LDS latencies
            for (int i0 = (int)W - 2; i0 >= 0; i0 -= 8) {
                uint32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = i0 - k >= 0 ? arr[(i0 - k) * PART_THREADS + tid] : 0u;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 - k;
                    if (i >= 0) {
                        const bool lt = v[k] < run;
                        run = lt ? v[k] : run;
                        arr[i * PART_THREADS + tid] = run;
                        if (!WRITE) { ri = lt ? (uint32_t)i : ri; sidx[i * PART_THREADS + tid] = (uint8_t)ri; }
                    }
                }
            }
            bi = 0;
        }
        // the suffix minimum the next position needs (index bi + 1 of the block before the current one)
        if (bi + 1 < W) sfx_next = arr[(bi + 1) * PART_THREADS + tid];
    }
    close_run();
    if (!WRITE) {
        // pack the runs: bits 0-3 their number (SUMMARY_OVERFLOW: too many, the read goes through the scanning
        // scatter), then 12 bits per run from bit 8
        const uint64_t lo = sum_lo | (qn <= (uint32_t)SUMMARY_RUNS ? qn : SUMMARY_OVERFLOW), hi = sum_hi;
        summaries[r] = uint4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
        if (read_classes) { read_classes[r] = cmask; if (uint64_t* rc_ = const_cast<uint64_t*>(run_classes_of(read_classes, n_reads))) rc_[r] = rcls; }
        // a read with more runs than a summary holds goes on the list of reads the scanning scatter handles (two in
        // 10^5; in the counting scan `bucket_cur` is that list's counter, `records` the list, `n_out` its capacity)
        if (qn > (uint32_t)SUMMARY_RUNS && bucket_cur) {
            const unsigned long long at = atomicAdd(bucket_cur, 1ull);
            if (at < n_out) reinterpret_cast<uint32_t*>(records)[at] = (uint32_t)r;
        }
    }
    if (WRITE) {
        for (uint32_t e = 0; e < qn; ++e) {
            uint32_t a = queue[(2 * e) * PART_THREADS + tid], b = queue[(2 * e + 1) * PART_THREADS + tid];
            const uint64_t dst = place(b);
            if (dst != ~0ull) emit_record<K>(words, n_words, bit0, a & 0xFFFFu, a >> 16, b, gl, tag, dst, records);
        }
    }
}

template <int K, bool WRITE>
__global__ void __launch_bounds__(PART_THREADS)
k_partition(const uint8_t* __restrict__ packed, uint64_t packed_bytes, const uint64_t* __restrict__ base_off,
            const uint32_t* __restrict__ good_len, const int32_t* __restrict__ bc, int64_t ign_bc_below,
            uint64_t n_reads, PartParams pp,
            unsigned long long* __restrict__ bucket_acc,      // !WRITE: per fine bucket (records<<32 | instances) ...
            unsigned long long* __restrict__ class_hist,      // ... or (sharded) per owner and class: [records | instances][world * PART_CLASSES]
            unsigned long long* __restrict__ bucket_cur,      // WRITE: append cursors, start at each fine bucket's first record index
            uint64_t n_out,                                   // WRITE: records the pass holds (nothing is written beyond)
            uint4* __restrict__ records,                      // (!WRITE: these three are the list of reads whose summary overflowed:
                                                              //  its counter, its capacity and -- as uint32_t* -- its entries)
            uint4* __restrict__ summaries,                    // !WRITE: per read, its runs (see SUMMARY_RUNS)
            const uint32_t* __restrict__ read_list,           // WRITE (optional): the reads to process
            uint64_t n_list,
            const uint64_t* __restrict__ slice_base,          // WRITE, sharded: records go to owner slices; bucket_cur = fill of each slice
            uint32_t* __restrict__ read_classes)              // !WRITE: per read, the classes of its runs' buckets (sweep_class_of)
{
    extern __shared__ uint32_t smem[];
    // class counts are gathered in LDS by a grid-stride launch (a few thousand blocks) and flushed once per
    // block: one global atomic per run on a handful of addresses would serialise
    uint32_t* lh = nullptr;
    const uint32_t n_bins = 2u * (PART_CLASSES << pp.log2_world);
    if (!WRITE && class_hist) {
        lh = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(smem) + 5 * pp.W * PART_THREADS + 4 * PART_RING * PART_THREADS);
        for (uint32_t i = threadIdx.x; i < n_bins; i += PART_THREADS) lh[i] = 0;
        __syncthreads();
    }
    const uint64_t n_work = (WRITE && read_list) ? n_list : n_reads;
    for (uint64_t r0 = (uint64_t)blockIdx.x * PART_THREADS; r0 < n_work; r0 += (uint64_t)gridDim.x * PART_THREADS)
        partition_read<K, WRITE>(r0 + threadIdx.x, smem, lh, packed, packed_bytes, base_off, good_len, bc, ign_bc_below, n_reads, pp,
                                 bucket_acc, bucket_cur, n_out, records, summaries, read_list, n_list, slice_base, read_classes);
    if (lh) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_bins; i += PART_THREADS) if (lh[i]) atomicAdd(&class_hist[i], (unsigned long long)lh[i]);
    }
}

// ---- the counting scan, window arrays in registers (M = 16: the default minimizer length)
// Same result as k_partition<K, false> -- bucket counters (or class counts), run summaries, class masks, the list of
// reads with too many runs -- with the block prefix / suffix minima of the sliding window kept in W + W registers per lane
// instead of W x 5 bytes of LDS: the loop over a block of W m-mer positions is unrolled so that every index is a
// compile-time constant.  What that buys: no LDS operation per base (there were six, with their address arithmetic), and
// a run is closed by two register moves and an LDS store -- its bucket counter is bumped after the read's last base, in
// a loop all lanes walk together -- where the per-base version paid the whole close (atomic, class mask, summary field)
// as a divergent block in almost every step: among 64 reads some run ends nearly everywhere.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int K, int M>
__global__ void __launch_bounds__(PART_THREADS)
k_scan_count(const uint8_t* __restrict__ packed, uint64_t packed_bytes, const uint64_t* __restrict__ base_off,
             const uint32_t* __restrict__ good_len, uint64_t r_first, uint64_t n_reads /* reads [r_first, n_reads) */, PartParams pp,
             unsigned long long* __restrict__ bucket_acc, unsigned long long* __restrict__ class_hist,
             unsigned long long* __restrict__ ovf_count, uint64_t ovf_cap, uint32_t* __restrict__ ovf_list,
             uint4* __restrict__ summaries, uint32_t* __restrict__ read_classes)
{
    constexpr int W = K - M + 1;
    constexpr uint32_t mmask = M == 16 ? 0xFFFFFFFFu : ((1u << (2 * M)) - 1u);
    constexpr uint32_t rsh = 2 * (M - 1);
    constexpr int RUNS = SUMMARY_RUNS;                                   // buckets a lane remembers per read: as many as a summary holds
    extern __shared__ uint32_t smem[];
    uint32_t* ring = smem;                                               // [PART_RING][PART_THREADS] words of the read
    uint32_t* runb = smem + PART_RING * PART_THREADS;                    // [RUNS][PART_THREADS] bucket of each closed run
    uint32_t* lh = nullptr;
    const uint32_t n_bins = 2u * (PART_CLASSES << pp.log2_world);
    if (class_hist) {
        lh = runb + RUNS * PART_THREADS;
        for (uint32_t i = threadIdx.x; i < n_bins; i += PART_THREADS) lh[i] = 0;
        __syncthreads();
    }
    const int tid = threadIdx.x;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(packed);
    const uint64_t n_words = (packed_bytes + 3) >> 2;
    const uint32_t log2_local = pp.log2_nb - pp.log2_world;
    for (uint64_t r0 = r_first + (uint64_t)blockIdx.x * PART_THREADS; r0 < n_reads; r0 += (uint64_t)gridDim.x * PART_THREADS) {
        const uint64_t r = r0 + tid;
        const uint32_t gl = r < n_reads ? good_len[r] : 0;
        const bool live = gl >= (uint32_t)K + 1;                         // Kmerizer::map: len < K+1 emits nothing (:153)
        uint32_t qn = 0, cmask = 0;
        uint64_t sum_lo = 0, sum_hi = 0, rcls = 0;
        if (live) {
            const uint64_t bit0 = base_off[r] * 8;
            uint64_t wi = bit0 >> 5, filled_to = wi;
            auto fill = [&]() {
                uint32_t w[PART_RING];
#pragma unroll
                for (int k = 0; k < PART_RING; ++k) w[k] = filled_to + k < n_words ? words[filled_to + k] : 0u;
#pragma unroll
                for (int k = 0; k < PART_RING; ++k) ring[((filled_to + k) & (PART_RING - 1)) * PART_THREADS + tid] = w[k];
                filled_to += PART_RING;
            };
            fill();
            uint32_t wbits = ring[(wi & (PART_RING - 1)) * PART_THREADS + tid], wpos = (uint32_t)bit0 & 31u;
            auto next_base = [&]() -> uint32_t {
                const uint32_t b = (wbits >> wpos) & 3u;
                wpos += 2;
                if (wpos == 32) { wpos = 0; ++wi; if (wi == filled_to) fill(); wbits = ring[(wi & (PART_RING - 1)) * PART_THREADS + tid]; }
                return b;
            };
            uint32_t f = 0, rc = 0;
            for (int j = 0; j < M - 1; ++j) { const uint32_t b = next_base(); f = ((f << 2) | b) & mmask; rc = (rc >> 2) | ((3u - b) << rsh); }
            const uint32_t n_mmers = gl - M + 1;                         // m-mer positions t = 0 .. n_mmers-1; k-mer s = t-W+1 is complete at t >= W-1
            // suffix minima of the previous block, overwritten by the hashes of this one -- and where they sit, a BYTE each, four
            // to a register (every index is a constant once the block loop is unrolled: a field extract / insert).  With a word
            // per position the two arrays are 2W registers: 66 at K=48, and 90 at K=60, which the compiler kept in scratch --
            // 593 ms against the LDS-window scan's 430 (round 3); packed they are W + W/4 = 57.
            uint32_t arr[W], sidxw[(W + 3) / 4];
            auto sidx_get = [&](int i) -> uint32_t { return (sidxw[i >> 2] >> (8 * (i & 3))) & 255u; };
            auto sidx_set = [&](int i, uint32_t v) { const uint32_t sh = 8u * (uint32_t)(i & 3); sidxw[i >> 2] = (sidxw[i >> 2] & ~(255u << sh)) | (v << sh); };
#pragma unroll
            for (int i = 0; i < (W + 3) / 4; ++i) sidxw[i] = 0;
            uint32_t P = 0, Pi = 0;
            uint32_t cur_b = 0, cur_nk = 0, cur_rel = 0;
            auto close_run = [&]() {                                     // (cheap on purpose: see the head of this kernel)
                if (qn < (uint32_t)RUNS) {
                    runb[qn * PART_THREADS + tid] = cur_b;
                    const uint64_t fld = cur_nk | (cur_rel << 6);
                    const uint32_t b = 8u + 12u * qn;
                    if (b < 64u) { sum_lo |= fld << b; if (b > 52u) sum_hi |= fld >> (64u - b); } else sum_hi |= fld << (b - 64u);
                } else {
                    // the eleventh run and beyond (two reads in 10^5 have them; the read then goes through the scanning scatter):
                    // counted on the spot
                    if (lh) { const uint32_t bin = class_bin(cur_b, pp); atomicAdd(&lh[bin], 1u); atomicAdd(&lh[(PART_CLASSES << pp.log2_world) + bin], cur_nk); }
                    else atomicAdd(&bucket_acc[cur_b], (1ull << 32) | cur_nk);
                    cmask |= 1u << sweep_class_of(cur_b & ((1u << log2_local) - 1u), log2_local);
                }
                ++qn;
            };
            for (uint32_t t0 = 0; t0 < n_mmers; t0 += W) {              // one block of W m-mer positions; t = t0 + bi
                // (expanded by template recursion, not `#pragma unroll`: at W = 45 the optimizer declines the pragma and arr[] goes
                // to scratch; every index below must be a constant for arr[] / sidxw[] to be registers)
                static_for<0, W>([&](auto bi_c) {
                    constexpr int bi = decltype(bi_c)::value;
                    if (t0 + (uint32_t)bi < n_mmers) {
                        const uint32_t b = next_base();
                        f = ((f << 2) | b) & mmask;
                        rc = (rc >> 2) | ((3u - b) << rsh);
                        const uint32_t h = mmer_hash(f < rc ? f : rc);
                        const bool newmin = bi == 0 || h < P;
                        P = newmin ? h : P;
                        Pi = newmin ? (uint32_t)bi : Pi;
                        const bool first = t0 == 0;                      // (with bi == W-1: the read's first k-mer)
                        if (!first || bi == W - 1) {                      // a k-mer ends here
                            uint32_t mv = P;
                            bool older = false;
                            if constexpr (bi != W - 1) { older = arr[bi + 1] < mv; mv = older ? arr[bi + 1] : mv; }
                            const uint32_t bucket = bucket_of(mv, pp);
                            const bool open = first || bucket != cur_b || cur_nk == (uint32_t)KTraits<K>::NK_MAX;
                            if (open) {
                                if (!first) close_run();
                                cur_b = bucket; cur_nk = 1;
                                cur_rel = (bi != W - 1 && older) ? sidx_get(bi + 1 < W ? bi + 1 : 0) - (uint32_t)bi - 1u : (uint32_t)(W - 1 - bi) + Pi;
                            } else ++cur_nk;
                        }
                        arr[bi] = h;
                        if (bi == W - 1) {                                // block complete: its hashes become suffix minima
                            uint32_t run = h, ri = W - 1;
                            sidx_set(W - 1, W - 1);
                            static_for<0, W - 1>([&](auto j_c) {
                                constexpr int i = W - 2 - decltype(j_c)::value;
                                const bool lt = arr[i] < run; run = lt ? arr[i] : run; arr[i] = run; ri = lt ? (uint32_t)i : ri; sidx_set(i, ri);
                            });
                        }
                    }
                });
            }
            close_run();
        }
        // ---- the runs' buckets: counters and class mask, all lanes together (a read has ~4 runs)
        const uint32_t mine = min(qn, (uint32_t)RUNS);
        for (uint32_t i = 0; __ballot(i < mine) != 0ull; ++i) {
            if (i < mine) {
                const uint32_t cb = runb[i * PART_THREADS + tid];
                const uint32_t bfld = 8u + 12u * i;
                const uint32_t nk = (uint32_t)(bfld + 12 <= 64 ? sum_lo >> bfld : bfld >= 64 ? sum_hi >> (bfld - 64) : (sum_lo >> bfld) | (sum_hi << (64 - bfld))) & 63u;
                if (lh) { const uint32_t bin = class_bin(cb, pp); atomicAdd(&lh[bin], 1u); atomicAdd(&lh[(PART_CLASSES << pp.log2_world) + bin], nk); }
                else atomicAdd(&bucket_acc[cb], (1ull << 32) | nk);
                const uint32_t cls = sweep_class_of(cb & ((1u << log2_local) - 1u), log2_local);
                cmask |= 1u << cls;
                rcls |= (uint64_t)cls << (5u * i);
            }
        }
        if (r < n_reads) {
            const uint64_t lo = sum_lo | (qn <= (uint32_t)SUMMARY_RUNS ? qn : SUMMARY_OVERFLOW);
            summaries[r] = uint4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)sum_hi, (uint32_t)(sum_hi >> 32)};
            if (read_classes) { read_classes[r] = cmask; if (uint64_t* rc_ = const_cast<uint64_t*>(run_classes_of(read_classes, n_reads))) rc_[r] = rcls; }
        }
        if (qn > (uint32_t)SUMMARY_RUNS) {
            const unsigned long long at = atomicAdd(ovf_count, 1ull);
            if (at < ovf_cap) ovf_list[at] = (uint32_t)r;
        }
    }
    if (lh) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_bins; i += PART_THREADS) if (lh[i]) atomicAdd(&class_hist[i], (unsigned long long)lh[i]);
    }
}

// Sharded scatter: the records of one pass go into one slice per owner rank, in no particular order inside
// the slice (the owner regroups them by the bucket id in the header).  A block handles SLICE_READS reads per
// thread: first it counts its records per owner (LDS), reserves room in every slice with one global atomic
// per owner, then writes.
constexpr int SLICE_READS = 16;
template <int K, bool EMIT, typename F>
__device__ __forceinline__ void for_each_run_in_pass(uint64_t r, const uint8_t* __restrict__ packed, uint64_t packed_bytes,
                                                     const uint64_t* __restrict__ base_off, const PartParams& pp,
                                                     const uint4* __restrict__ summaries, const uint64_t* __restrict__ run_classes, F&& f)
{
    const uint4 sm = summaries[r];
    const uint32_t n = sm.x & 15u;
    if (n == 0 || n == SUMMARY_OVERFLOW) return;
    const uint64_t rcls = run_classes ? run_classes[r] : 0ull;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(packed);
    const uint64_t n_words = (packed_bytes + 3) >> 2;
    const uint64_t bit0 = base_off[r] * 8;
    const uint32_t M = pp.M;
    const uint32_t mmask = M == 16 ? 0xFFFFFFFFu : ((1u << (2 * M)) - 1u);
    const uint64_t lo = (uint64_t)sm.x | ((uint64_t)sm.y << 32), hi = (uint64_t)sm.z | ((uint64_t)sm.w << 32);
    uint32_t s0 = 0;
#pragma unroll
    for (int i = 0; i < SUMMARY_RUNS; ++i) {
        if ((uint32_t)i < n) {
            constexpr int B0 = 8;
            const int b = B0 + 12 * i;
            const uint32_t fld = (uint32_t)(b + 12 <= 64 ? lo >> b : b >= 64 ? hi >> (b - 64) : (lo >> b) | (hi << (64 - b))) & 0xFFFu;
            const uint32_t nk = fld & 63u, rel = fld >> 6;
            if (run_classes && !((pp.class_mask >> ((uint32_t)(rcls >> (5 * i)) & 31u)) & 1u)) { s0 += nk; continue; }   // a run of a class this pass does not touch
            const uint64_t bo = bit0 + 2ull * (s0 + rel);
            const uint64_t wi = bo >> 5;
            const uint32_t w0 = words[wi], w1 = wi + 1 < n_words ? words[wi + 1] : 0u;
            const uint32_t x = alignbit(w1, w0, (uint32_t)bo & 31u) & mmask;     // the m-mer, first base in the low bits
            uint32_t y = __brev(x);
            y = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
            const uint32_t fw = y >> (32 - 2 * M), rcv = ~x & mmask;             // the scan's forward / reverse-complement values
            const uint32_t bucket = bucket_of(mmer_hash(fw < rcv ? fw : rcv), pp);
            const uint32_t lb = pass_local(bucket, pp);
            if (lb != 0xFFFFFFFFu) f(s0, nk, lb, bucket >> (pp.log2_nb - pp.log2_world), bit0, (uint32_t)i + 1 == n);
            s0 += nk;
        }
    }
}

template <int K>
__global__ void __launch_bounds__(256)
k_scatter_slices(const uint8_t* __restrict__ packed, uint64_t packed_bytes, const uint64_t* __restrict__ base_off,
                 const uint32_t* __restrict__ good_len, const int32_t* __restrict__ bc, int64_t ign_bc_below,
                 uint64_t n_reads, PartParams pp, const uint4* __restrict__ summaries, const uint32_t* __restrict__ read_classes,
                 const uint64_t* __restrict__ slice_base,        // [world + 1] first record index of every owner's slice
                 unsigned long long* __restrict__ slice_fill,    // [world] records reserved so far in every slice
                 uint4* __restrict__ records)
{
    __shared__ uint32_t cnt[256];                                 // per owner: records of this block, then its write cursor
    __shared__ unsigned long long at[256];
    __shared__ uint16_t list[256 * SLICE_READS];                  // the block's reads that can have a run in this pass (class masks)
    __shared__ uint32_t n_list;
    const uint32_t world = 1u << pp.log2_world;
    const int lane = threadIdx.x & 63;
    const uint64_t r_block = (uint64_t)blockIdx.x * 256 * SLICE_READS;
    if (threadIdx.x < world) cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) n_list = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SLICE_READS; ++j) {
        const uint32_t rel = 256u * j + threadIdx.x;
        const uint64_t r = r_block + rel;
        const bool hit = r < n_reads && (!read_classes || (read_classes[r] & pp.class_mask) != 0u);
        const unsigned long long mk = __ballot(hit);
        uint32_t w = 0;
        if (lane == 0 && mk) w = atomicAdd(&n_list, (uint32_t)__popcll(mk));
        w = __builtin_amdgcn_readfirstlane(w);
        if (hit) list[w + __popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)rel;
    }
    __syncthreads();
    const uint32_t n = n_list;
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        for_each_run_in_pass<K, false>(r_block + list[i], packed, packed_bytes, base_off, pp, summaries, run_classes_of(read_classes, n_reads),
                                       [&](uint32_t, uint32_t, uint32_t, uint32_t owner, uint64_t, bool) { atomicAdd(&cnt[owner], 1u); });
    __syncthreads();
    if (threadIdx.x < world) {
        const uint32_t m = cnt[threadIdx.x];
        at[threadIdx.x] = slice_base[threadIdx.x] + (m ? atomicAdd(&slice_fill[threadIdx.x], (unsigned long long)m) : 0ull);
        cnt[threadIdx.x] = 0;
    }
    __syncthreads();
    const uint32_t* words = reinterpret_cast<const uint32_t*>(packed);
    const uint64_t n_words = (packed_bytes + 3) >> 2;
    (void)good_len;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint64_t r = r_block + list[i];
        int32_t tag = -1;
        if (bc && (int64_t)r + pp.read_id0 >= ign_bc_below) tag = bc[r];
        for_each_run_in_pass<K, true>(r, packed, packed_bytes, base_off, pp, summaries, run_classes_of(read_classes, n_reads),
            [&](uint32_t s0, uint32_t nk, uint32_t lb, uint32_t owner, uint64_t bit0, bool last) {
                const uint64_t dst = at[owner] + atomicAdd(&cnt[owner], 1u);
                // only the read's last run has no successor base (its last k-mer ends the trimmed read)
                if (dst < slice_base[owner + 1]) emit_record<K>(words, n_words, bit0, s0, nk, lb, last ? 0u : 0xFFFFFFFFu, tag, dst, records);
            });
    }
}

// One pass of the scatter, from the run summaries: a lane walks its read's runs, rebuilds each
// run's bucket from the minimizer the summary points at (2M bits of the read, no scan), and writes the
// records of the runs that belong to this pass.  The cursors start at the buckets' first record indices;
// afterwards cursor[b] must equal base[b+1] (k_check_cursors).
// reads per thread of a k_scatter_runs block.  The block's list lives in LDS, and the sweep runs beside k_count,
// whose two workgroups per CU leave 16 KB of LDS at K=40/48 and 8 KB at K=60: with 4 KB lists a K=60 sweep got one
// block per CU and ran 3.6 times longer than alone.
#ifndef DFK_SWEEP_READS
#define DFK_SWEEP_READS 8
#endif
template <int K> constexpr int sweep_reads() { return KTraits<K>::KW == 4 ? 2 : DFK_SWEEP_READS; }
template <int K>
__global__ void __launch_bounds__(256)
k_scatter_runs(const uint8_t* __restrict__ packed, uint64_t packed_bytes, const uint64_t* __restrict__ base_off,
               const uint32_t* __restrict__ good_len, const int32_t* __restrict__ bc, int64_t ign_bc_below,
               uint64_t n_reads, PartParams pp, const uint4* __restrict__ summaries, const uint32_t* __restrict__ read_classes,
               unsigned long long* __restrict__ bucket_cur, uint64_t n_out, uint4* __restrict__ records)
{
    // A block owns 256 * SWEEP_READS consecutive reads.  First the class masks (4 bytes per read, coalesced)
    // pick the reads that can have a run in this pass; their indices are packed into an LDS list, which the
    // threads then work through densely (lanes whose read is not in the pass would otherwise idle through the
    // latency chain summary -> bases -> cursor -> record of their neighbours).
    constexpr int SWEEP_READS = sweep_reads<K>();
    __shared__ uint16_t list[256 * SWEEP_READS];
    __shared__ uint32_t n_list;
    (void)good_len;
    const int lane = threadIdx.x & 63;
    const uint64_t r_block = (uint64_t)blockIdx.x * 256 * SWEEP_READS;
    if (threadIdx.x == 0) n_list = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SWEEP_READS; ++j) {
        const uint32_t rel = 256u * j + threadIdx.x;
        const uint64_t r = r_block + rel;
        const bool hit = r < n_reads && (read_classes[r] & pp.class_mask) != 0u;
        const unsigned long long mk = __ballot(hit);
        uint32_t at = 0;
        if (lane == 0 && mk) at = atomicAdd(&n_list, (uint32_t)__popcll(mk));
        at = __builtin_amdgcn_readfirstlane(at);
        if (hit) list[at + __popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)rel;
    }
    __syncthreads();
    const uint32_t n = n_list;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(packed);
    const uint64_t n_words = (packed_bytes + 3) >> 2;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint64_t r = r_block + list[i];
        int32_t tag = -1;
        if (bc && (int64_t)r + pp.read_id0 >= ign_bc_below) tag = bc[r];
        for_each_run_in_pass<K, true>(r, packed, packed_bytes, base_off, pp, summaries, run_classes_of(read_classes, n_reads),
            [&](uint32_t s0, uint32_t nk, uint32_t lb, uint32_t, uint64_t bit0, bool last) {
                const uint64_t dst = atomicAdd(&bucket_cur[lb], 1ull);             // one random access for base and rank
                // only the read's last run has no successor base (its last k-mer ends the trimmed read)
                if (dst < n_out) emit_record<K>(words, n_words, bit0, s0, nk, lb, last ? 0u : 0xFFFFFFFFu, tag, dst, records);
            });
    }
}

// ============================================================================ a2 (second half) + a3 + a4 + a5: count
// A work item is a contiguous range of records.  One workgroup counts one item at a time in
// an open-addressing table: KW key words + count + context + barcode word per slot, SoA.
// Each wave takes 64 records at a time, stages them in its private LDS area, maps the
// item's k-mer instances densely onto lanes (bitmask of run starts + popcount rank), rebuilds
// each instance's canonical k-mer and context from the staged 2-bit stream and inserts it.
// After the item, solid slots are compacted to the output and the spectrum is updated.


struct ItemRange { uint32_t b0, b1; };     // a work item: fine buckets [b0,b1) of the current pass

struct CountParams {
    uint32_t min_freq, min_bc;
    uint32_t n_items;
    uint32_t single;                 // every record of this launch holds ONE k-mer (the sub-buckets of split hot buckets): no staging
    uint64_t seg_cap;                // entries the output buffer holds (k_count: the part's reservation; k_big_emit: the fallback buffer)
    uint32_t do_adj;                 // resolve adjacencies inside the table where possible (min_freq > 1)
    uint32_t keep_pre;               // keep the pre-adjacency context byte in the entry's pad field (tests)
};

struct CountGlobals {                // device-resident counters
    unsigned long long n_distinct;
    unsigned long long big_cursor;   // entries written to the fallback buffer by k_big_emit
    unsigned int next_item;
    unsigned int n_overflow;         // items that overflowed their table
    unsigned int solid_overflow;     // the output buffer ran out of room
    unsigned int n_split;            // items that overflowed their table and were cut in two by their workgroup
    unsigned long long n_boundary;   // solid entries left with unresolved (cross-item) context bits
    unsigned long long part_cursor;  // entries of the part's reservation handed out to workgroups so far (whole chunks)
    unsigned int n_wg_idle;          // workgroups of the last launch that ended without having counted an item (placed late, or never needed)
    unsigned int pad_idle;
};

// Where a persistent workgroup is in the chunk of the output buffer it is filling (kept across the launches of
// one pass).  Workgroups take chunks of OUT_CHUNK entries from CountGlobals::part_cursor and fill them item by
// item, an item's entries running over into a fresh chunk when needed; only the last chunk of every workgroup
// is left partly empty, and the host moves entries from the tail into those holes.
struct WgOut { unsigned long long chunk; unsigned int used; unsigned int pad; };

constexpr int COUNT_HIST_BINS = 256;             // spectrum bins kept in LDS; higher counts go straight to the global bins
constexpr uint32_t COUNT_MAX_PROBE = 96;
constexpr uint32_t HIST_GLOBAL_BINS = 1u << 24;   // KDef count saturates at 2^24-1 (ReadPather.h:128-129)
constexpr int COUNT_CHUNK = 32;                   // records a wave stages at a time
constexpr uint32_t FLAG_SOLID = 0x80000000u;      // barcode word reused after counting: solid flag | unresolved context bits

template <int K> struct WaveStage {               // per-wave private LDS
    uint32_t rec[COUNT_CHUNK * 8 + 8];            // staged records (+ pad for the 5-word window read)
    uint32_t starts[COUNT_CHUNK];                 // first instance index of each record
    uint32_t msk[64];                             // bit t set <=> instance t starts a record (<= 32*55 bits)
    uint32_t pc[64];                              // exclusive popcount prefix of msk words
};

// Table words are read and written with relaxed agent-scope atomics: for the LDS table these
// are plain ds_read/ds_write; for the HBM fallback table they bypass the CU's L1, which other
// waves' stores and atomics do not update (MI355X_MICROARCH.md, inter-workgroup visibility).
__device__ __forceinline__ uint32_t tld(const uint32_t* p)
{ return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tst(uint32_t* p, uint32_t v)
{ __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// State of one key being inserted.  Slot state lives in the count word (see CNT_LOCK).  The 8-bit
// fingerprint lets a probe skip a foreign slot on the value its CAS returned, without reading the key
// words; the probe sequence is double hashing (odd step from the hash), which keeps the longest probe
// of a wave short at 75 % load.
struct Probe {
    uint32_t k0, k1, k2, k3, ctx, fp, step, slot;
    int32_t tag;
    bool active;
};

__device__ __forceinline__ Probe probe_begin(u128 c, uint32_t ctx, int32_t tag, uint32_t S, bool active)
{
    const uint32_t h = key_hash(c);
    return Probe{(uint32_t)c.lo, (uint32_t)(c.lo >> 32), (uint32_t)c.hi, (uint32_t)(c.hi >> 32), ctx, (h >> 24) | 1u,
                 ((h >> 11) | 1u) & (S - 1), h & (S - 1), tag, active};
}

// Insert one key per lane: find or claim its slot (divergent loop), then count it (convergent).
//
// Loop body, per probe: CAS(cnt[slot], 0 -> CNT_LOCK) and, for the LDS table, the slot's key words in the
// same batch of LDS operations -- the hardware runs a wave's LDS operations in order, so the key reads
// see everything the slot's owner wrote before the count word the CAS returned (one round trip per probe
// instead of two).  For the HBM table the reads follow the acquire.  A claimed slot gets its key words and
// is published with count 0; its context and barcode words are zero already (every finish leaves the
// table zeroed), so claimers and matchers do the same thing afterwards.
//
// The loop condition is wave-uniform (ballot) on purpose.  A lane that finds CNT_LOCK waits for the lane
// initialising the slot, which may sit in the same wave.  With a per-lane `while (!done)` the compiler may
// turn the winner's branch (it ends in the loop exit) into an exit block; the wave then runs it only after
// every lane has left the loop and the waiters spin forever.  With the exit decided only at the header, the
// winner's stores are inside the loop body.  Every spin is bounded, so every wave reaches the end of the
// kernel whatever happens.  Returns false if the probe sequence got too long.
// (Two keys per lane in flight was tried and was slower.)
#define DFK_COMPILER_FENCE() asm volatile("" ::: "memory")
#ifdef DFK_PHASE_TIMES    // experiment only: where a k_count wave's cycles go (s_memtime deltas summed over all waves)
__device__ unsigned long long g_phase[16];
struct PhaseClock {
    unsigned long long t, acc[12];
    __device__ __forceinline__ void start() { t = __builtin_amdgcn_s_memtime(); for (int i = 0; i < 12; ++i) acc[i] = 0; }
    __device__ __forceinline__ void mark(int i) { const unsigned long long n = __builtin_amdgcn_s_memtime(); acc[i] += n - t; t = n; }
    __device__ __forceinline__ void flush(int lane) { if (lane == 0) for (int i = 0; i < 12; ++i) atomicAdd(&g_phase[i], acc[i]); }
};
#define PH(i) phase.mark(i)
#else
#define PH(i) do { } while (0)
#endif
#ifdef DFK_PROBE_STATS   // experiment only: [0] loop iterations, [1] batches, [2] lane probes, [3] waits on a locked slot
__device__ unsigned long long g_probe_stats[4];
#endif

// what the first insert that gave up in an HBM table looked like: {set, probe steps, waits on a locked slot, log2 slots}
__device__ unsigned int g_big_fail[4];

// lane state in the probe loop
enum : uint32_t { PS_PROBING = 0, PS_FOUND = 1, PS_FAILED = 2, PS_IDLE = 3 };
// one counter bounds both the probe sequence (COUNT_MAX_PROBE steps) and the waits on a locked slot
constexpr uint32_t PROBE_COST = 1u << 12, PROBE_LIMIT = (COUNT_MAX_PROBE + 1) * PROBE_COST;

template <int KW, int NBC, bool LDS_TABLE>
__device__ __forceinline__ bool table_insert(uint32_t* __restrict__ keys, uint32_t* __restrict__ cnt,
                                             uint32_t* __restrict__ ctxs, uint32_t* __restrict__ bcw,
                                             uint32_t S, const Probe& A, uint32_t& n_claimed)
{
    uint32_t state = A.active ? PS_PROBING : PS_IDLE, slot = A.slot, seen = 0, cost = 0, waits = 0;
#ifdef DFK_PROBE_STATS
    uint32_t dbg_iter = 0;
#endif
    while (__ballot(state == PS_PROBING) != 0ull) {
#ifdef DFK_PROBE_STATS
        ++dbg_iter;
#endif
        if (state == PS_PROBING) {
            uint32_t e = 0, r0, r1, r2, r3 = A.k3;
            if (LDS_TABLE) {
                __hip_atomic_compare_exchange_strong(&cnt[slot], &e, CNT_LOCK, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP);
                DFK_COMPILER_FENCE();
                r0 = tld(&keys[slot]); r1 = tld(&keys[S + slot]); r2 = tld(&keys[2 * S + slot]);
                if (KW == 4) r3 = tld(&keys[3 * S + slot]);
            } else {
                __hip_atomic_compare_exchange_strong(&cnt[slot], &e, CNT_LOCK, __ATOMIC_ACQUIRE, __ATOMIC_ACQUIRE,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                r0 = tld(&keys[slot]); r1 = tld(&keys[S + slot]); r2 = tld(&keys[2 * S + slot]);
                if (KW == 4) r3 = tld(&keys[3 * S + slot]);
            }
            const bool won = e == 0u;                                    // the slot was empty and is ours now
            if (won) {
                tst(&keys[slot], A.k0); tst(&keys[S + slot], A.k1); tst(&keys[2 * S + slot], A.k2);
                if (KW == 4) tst(&keys[3 * S + slot], A.k3);
                if (LDS_TABLE) {
                    DFK_COMPILER_FENCE();
                    __hip_atomic_store(&cnt[slot], A.fp << 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else __hip_atomic_store(&cnt[slot], A.fp << 24, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            // the rest is selects, no branches.  A locked slot (fingerprint 0) or an empty one never matches.
            // (written as arithmetic on 0/1 words so that the compiler does not turn it back into control flow)
            const uint32_t locked = e == CNT_LOCK;
            const uint32_t match = ((e >> 24) == A.fp) & (r0 == A.k0) & (r1 == A.k1) & (r2 == A.k2) & (r3 == A.k3);
            const uint32_t fin = (uint32_t)won | match;
            n_claimed += (uint32_t)won;
            seen |= e & (0u - match);                                    // 0 for the claimer
            cost += PROBE_COST - (PROBE_COST - 1u) * locked;
            if (!LDS_TABLE) waits += locked;                             // (diagnostics of a failure only)
            slot = (slot + (A.step & ((fin | locked) - 1u))) & (S - 1);  // stay on a found or locked slot
            state = fin | ((uint32_t)(cost >= PROBE_LIMIT) << 1);        // PS_FOUND, PS_FAILED or PS_PROBING
        }
    }
#ifdef DFK_PROBE_STATS
    {
        uint32_t pr = cost / PROBE_COST, sp = cost % PROBE_COST;
        for (int d = 32; d > 0; d >>= 1) { pr += __shfl_down(pr, d, 64); sp += __shfl_down(sp, d, 64); }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&g_probe_stats[0], (unsigned long long)dbg_iter); atomicAdd(&g_probe_stats[1], 1ull);
            atomicAdd(&g_probe_stats[2], (unsigned long long)pr); atomicAdd(&g_probe_stats[3], (unsigned long long)sp);
        }
    }
#endif
    if (!LDS_TABLE && state == PS_FAILED && atomicCAS(&g_big_fail[0], 0u, 1u) == 0u) {
        g_big_fail[1] = (cost - waits) / PROBE_COST; g_big_fail[2] = waits; g_big_fail[3] = 31u - (uint32_t)__clz(S);
    }
    if (state == PS_FOUND) {
        if (LDS_TABLE) {
            // Counts saturate at exactly 2^24-1 (KDef::setCount).  The count shares its word with the fingerprint, so an
            // add must never carry.  Below 2^23 (as seen at probe time) a plain add is safe: between a lane's probe and
            // its add the other waves of the workgroup can add at most a few hundred thousand -- the probe loop is
            // bounded by PROBE_LIMIT iterations.  (The margin used to be 255 and "an item holds a few thousand
            // instances" the argument; but ONE fine bucket is one item however large: 330 k poly-A reads put 1.76x10^7
            // instances of one k-mer through one table, the count carried into the fingerprint and the k-mer claimed a
            // second slot -- in two runs of six.)  From 2^23 on the lanes of a wave that hit the same slot are added
            // as ONE compare-and-swap by their leader, which saturates exactly.
            const bool high = (seen & CNT_MASK) >= CNT_HALF;
            if (!high) atomicAdd(&cnt[slot], 1u);
            unsigned long long todo = __ballot(high);
            while (todo) {                                                    // (wave-uniform: `todo` is a ballot)
                const int leader = __ffsll((long long)todo) - 1;
                const uint32_t lslot = (uint32_t)__builtin_amdgcn_readlane((int)slot, leader);
                const unsigned long long same = __ballot(high && slot == lslot) & todo;
                if ((threadIdx.x & 63) == leader) {
                    const uint32_t k = (uint32_t)__popcll(same);
                    uint32_t cur = tld(&cnt[slot]);
                    for (;;) {
                        const uint32_t c0 = cur & CNT_MASK, c1 = c0 + k < CNT_MASK ? c0 + k : CNT_MASK;
                        if (c1 == c0) break;
                        if (__hip_atomic_compare_exchange_strong(&cnt[slot], &cur, (cur & ~CNT_MASK) | c1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                    }
                }
                todo &= ~same;
            }
        } else {
            // An HBM table is shared by the whole grid, and a hot k-mer (poly-A: 10^7 instances of ONE key) sends every
            // resident lane to one slot.  No margin on the 24-bit field is safe there: between a lane's probe and its add
            // any number of other lanes can finish theirs, and a carry into the fingerprint byte makes the k-mer claim a
            // second slot (seen: 330 k poly-A reads gave one distinct k-mer too many).  So the count of an HBM table is a
            // word of its own (after the context words), clamped by undoing an add that finds it at 2^31; the readers
            // (k_big_flags, k_big_emit) take min(count, 2^24-1).
            uint32_t* wide = ctxs + S;
            if (atomicAdd(&wide[slot], 1u) >= 0x7FFFFFFFu) atomicSub(&wide[slot], 1u);
        }
        atomicOr(&ctxs[slot], A.ctx);
        if (NBC > 0) {
            // areEnoughBarcodes (BuildReadQGraph48.cc:112-132): the slot remembers up to NBC = max(1, MIN_BC-1)
            // distinct positive barcodes (word j of the slot: bcw[j*S + slot], 0 = free); BCW_MULTI in word 0 says
            // "passes": one barcode more than it can remember has arrived, or an ignored read (tag -1).  A barcode
            // tries the words in order and stops at itself or at the first free word, so it is never held twice.
            if (A.tag == -1) atomicOr(&bcw[slot], BCW_MULTI);
            else if (A.tag > 0) {
                bool settled = false;
#pragma unroll
                for (int j = 0; j < NBC; ++j)
                    if (!settled) {
                        const uint32_t old = atomicCAS(&bcw[(size_t)j * S + slot], 0u, (uint32_t)A.tag);
                        settled = old == 0u || (old & ~BCW_MULTI) == (uint32_t)A.tag || (j == 0 && (old & BCW_MULTI));
                    }
                if (!settled) atomicOr(&bcw[slot], BCW_MULTI);
            }
        }
    }
    return state != PS_FAILED;
}

// Two keys per lane, probed side by side (LDS tables only).  k_count's waves spend their time waiting, not issuing: a probe
// is an LDS round trip (compare-and-swap + the key words) that the next step depends on, a batch of 64 keys takes 3.2 of
// them in a row (the longest probe sequence of the wave), and at four waves per SIMD there is not enough else to run
// meanwhile -- ten or twelve waves per workgroup count 8 / 15 % faster with the same instructions, and
// taking two fifths of the extraction's instructions away (pair_second) changed nothing.  Here the round trips of two
// batches overlap: one loop iteration issues both keys' LDS operations before it looks at either answer.  Both keys of a
// lane may be the same k-mer (a homopolymer run) or hash to the same slot: the operations of a wave execute in order, so the
// second key's compare-and-swap sees the first one's lock and comes back in the next iteration, as a key of another lane
// would.
template <int KW, int NBC>
__device__ __forceinline__ bool table_insert2(uint32_t* __restrict__ keys, uint32_t* __restrict__ cnt,
                                              uint32_t* __restrict__ ctxs, uint32_t* __restrict__ bcw,
                                              uint32_t S, const Probe& A, const Probe& B, uint32_t& n_claimed)
{
    uint32_t sa = A.active ? PS_PROBING : PS_IDLE, sb = B.active ? PS_PROBING : PS_IDLE;
    uint32_t slot_a = A.slot, slot_b = B.slot, seen_a = 0, seen_b = 0, cost_a = 0, cost_b = 0;
    while (__ballot((sa == PS_PROBING) | (sb == PS_PROBING)) != 0ull) {
        uint32_t ea = 0, a0 = 0, a1 = 0, a2 = 0, a3 = A.k3, eb = 0, b0 = 0, b1 = 0, b2 = 0, b3 = B.k3;
        const bool pa = sa == PS_PROBING, pb = sb == PS_PROBING;
        if (pa) {
            __hip_atomic_compare_exchange_strong(&cnt[slot_a], &ea, CNT_LOCK, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            DFK_COMPILER_FENCE();
            a0 = tld(&keys[slot_a]); a1 = tld(&keys[S + slot_a]); a2 = tld(&keys[2 * S + slot_a]);
            if (KW == 4) a3 = tld(&keys[3 * S + slot_a]);
        }
        DFK_COMPILER_FENCE();
        if (pb) {
            __hip_atomic_compare_exchange_strong(&cnt[slot_b], &eb, CNT_LOCK, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            DFK_COMPILER_FENCE();
            b0 = tld(&keys[slot_b]); b1 = tld(&keys[S + slot_b]); b2 = tld(&keys[2 * S + slot_b]);
            if (KW == 4) b3 = tld(&keys[3 * S + slot_b]);
        }
        DFK_COMPILER_FENCE();
        if (pa) {
            const bool won = ea == 0u;
            if (won) {
                tst(&keys[slot_a], A.k0); tst(&keys[S + slot_a], A.k1); tst(&keys[2 * S + slot_a], A.k2);
                if (KW == 4) tst(&keys[3 * S + slot_a], A.k3);
                DFK_COMPILER_FENCE();
                __hip_atomic_store(&cnt[slot_a], A.fp << 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            const uint32_t locked = ea == CNT_LOCK;
            const uint32_t match = ((ea >> 24) == A.fp) & (a0 == A.k0) & (a1 == A.k1) & (a2 == A.k2) & (a3 == A.k3);
            const uint32_t fin = (uint32_t)won | match;
            n_claimed += (uint32_t)won;
            seen_a |= ea & (0u - match);
            cost_a += PROBE_COST - (PROBE_COST - 1u) * locked;
            slot_a = (slot_a + (A.step & ((fin | locked) - 1u))) & (S - 1);
            sa = fin | ((uint32_t)(cost_a >= PROBE_LIMIT) << 1);
        }
        DFK_COMPILER_FENCE();
        if (pb) {
            const bool won = eb == 0u;
            if (won) {
                tst(&keys[slot_b], B.k0); tst(&keys[S + slot_b], B.k1); tst(&keys[2 * S + slot_b], B.k2);
                if (KW == 4) tst(&keys[3 * S + slot_b], B.k3);
                DFK_COMPILER_FENCE();
                __hip_atomic_store(&cnt[slot_b], B.fp << 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            const uint32_t locked = eb == CNT_LOCK;
            const uint32_t match = ((eb >> 24) == B.fp) & (b0 == B.k0) & (b1 == B.k1) & (b2 == B.k2) & (b3 == B.k3);
            const uint32_t fin = (uint32_t)won | match;
            n_claimed += (uint32_t)won;
            seen_b |= eb & (0u - match);
            cost_b += PROBE_COST - (PROBE_COST - 1u) * locked;
            slot_b = (slot_b + (B.step & ((fin | locked) - 1u))) & (S - 1);
            sb = fin | ((uint32_t)(cost_b >= PROBE_LIMIT) << 1);
        }
    }
    // count, context, barcode: as table_insert (the comments on saturation and barcodes are there)
    auto found = [&](const Probe& P, uint32_t state, uint32_t slot, uint32_t seen) {
        const bool high = state == PS_FOUND && (seen & CNT_MASK) >= CNT_HALF;
        if (state == PS_FOUND && !high) atomicAdd(&cnt[slot], 1u);
        unsigned long long todo = __ballot(high);
        while (todo) {                                                        // (wave-uniform: `todo` is a ballot)
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lslot = (uint32_t)__builtin_amdgcn_readlane((int)slot, leader);
            const unsigned long long same = __ballot(high && slot == lslot) & todo;
            if ((int)(threadIdx.x & 63) == leader) {
                const uint32_t k = (uint32_t)__popcll(same);
                uint32_t cur = tld(&cnt[slot]);
                for (;;) {
                    const uint32_t c0 = cur & CNT_MASK, c1 = c0 + k < CNT_MASK ? c0 + k : CNT_MASK;
                    if (c1 == c0) break;
                    if (__hip_atomic_compare_exchange_strong(&cnt[slot], &cur, (cur & ~CNT_MASK) | c1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                }
            }
            todo &= ~same;
        }
        if (state != PS_FOUND) return;
        atomicOr(&ctxs[slot], P.ctx);
        if (NBC > 0) {
            if (P.tag == -1) atomicOr(&bcw[slot], BCW_MULTI);
            else if (P.tag > 0) {
                bool settled = false;
#pragma unroll
                for (int j = 0; j < NBC; ++j)
                    if (!settled) {
                        const uint32_t old = atomicCAS(&bcw[(size_t)j * S + slot], 0u, (uint32_t)P.tag);
                        settled = old == 0u || (old & ~BCW_MULTI) == (uint32_t)P.tag || (j == 0 && (old & BCW_MULTI));
                    }
                if (!settled) atomicOr(&bcw[slot], BCW_MULTI);
            }
        }
    };
    found(A, sa, slot_a, seen_a);
    found(B, sb, slot_b, seen_b);
    return sa != PS_FAILED && sb != PS_FAILED;
}

// One k-mer instance as read from the wave's staged chunk: header, barcode, the five payload words that
// cover it, and its index inside the record.
struct InstRegs { uint32_t hdr, tag, p0, p1, p2, p3, p4, q; };

// LDS reads only (three dependent levels: mask/prefix -> run start -> record words); issued one iteration
// ahead of the insertion that consumes them so that their latency overlaps the previous probe.
template <int K>
__device__ __forceinline__ InstRegs fetch_instance(const WaveStage<K>* __restrict__ st, uint32_t t)
{
    const uint32_t w = t >> 5;
    const uint32_t bits = tld(&st->msk[w]) & (0xFFFFFFFFu >> (31u - (t & 31u)));
    const uint32_t r = st->pc[w] + __popc(bits) - 1u;
    const uint32_t q = t - st->starts[r];
    const uint32_t* rec = st->rec + 8 * r;
    const uint32_t* p = rec + 2 + (q >> 4);
    return InstRegs{rec[0], rec[1], p[0], p[1], p[2], p[3], p[4], q};
}

// Rebuild the instance's canonical k-mer and context from the 2-bit stream.
template <int K>
__device__ __forceinline__ Probe make_probe(const InstRegs& in, uint32_t S, bool active)
{
    const uint32_t hdr = in.hdr, q = in.q;
    const uint32_t nk = hdr & 63u;
    // bits [2q, 2q + 2(K+2)) of the payload: pred, K bases, succ
    const uint32_t sh = (2u * q) & 31u;
    uint32_t x0 = alignbit(in.p1, in.p0, sh), x1 = alignbit(in.p2, in.p1, sh), x2 = alignbit(in.p3, in.p2, sh), x3 = alignbit(in.p4, in.p3, sh);
    u128 X{(uint64_t)x0 | ((uint64_t)x1 << 32), (uint64_t)x2 | ((uint64_t)x3 << 32)};
    const uint32_t pred = x0 & 3u;
    u128 ks = shr128(X, 2);
    constexpr int SB = 2 * K + 2;                                   // bit position of the successor base
    const uint32_t succ = (uint32_t)(SB >= 64 ? (X.hi >> (SB - 64)) : (X.lo >> SB)) & 3u;
    bool rev;
    u128 c = canonical<K>(ks, &rev);
    // KMerContext(pred,succ) / initialContext / finalContext (kmers/KMerContext.h:27-28,91-95)
    uint32_t ctx = 0;
    if (q > 0 || (hdr & 64u)) ctx |= 0x10u << pred;
    if (q + 1 < nk || (hdr & 128u)) ctx |= 1u << succ;
    if (rev) ctx = ctx_rc(ctx);
    return probe_begin(c, ctx, (int32_t)in.tag, S, active);
}

// Stage records [rb, min(rb+COUNT_CHUNK, re)) in the calling wave's LDS area and index their instances: a bit at each
// record's first instance index, popcount prefixes of the mask words (fetch_instance finds instance t's record by rank).
// Returns the number of instances staged (wave-uniform).  Lanes load half a record each (1 KiB per wave, coalesced).
template <int K>
__device__ __forceinline__ uint32_t wave_stage_piece(const uint4* __restrict__ records, uint64_t rb, uint64_t re,
                                                     WaveStage<K>* __restrict__ st, int lane)
{
    static_assert(COUNT_CHUNK == 32, "one uint4 per lane");
    const uint64_t hidx = 2 * rb + lane;
    uint4 v{0, 0, 0, 0};
    if (hidx < 2 * re) v = records[hidx];
    reinterpret_cast<uint4*>(st->rec)[lane] = v;
    st->msk[lane] = 0;
    wave_sync();
    const uint32_t nk = lane < COUNT_CHUNK ? (st->rec[8 * lane] & 63u) : 0u;
    const uint32_t incl = wave_incl_scan(nk, lane);
    const uint32_t start = incl - nk;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (lane < COUNT_CHUNK) st->starts[lane] = start;
    if (nk) atomicOr(&st->msk[start >> 5], 1u << (start & 31u));
    wave_sync();
    const uint32_t c0 = __popc(tld(&st->msk[lane]));
    st->pc[lane] = wave_incl_scan(c0, lane) - c0;
    wave_sync();
    return total;
}

// Count records [rb, min(rb+COUNT_CHUNK, re)) with the calling wave (wave-synchronous; no
// block barriers).
template <int K, int NBC, bool LDS_TABLE, bool SUB = false>
__device__ __forceinline__ uint32_t wave_count_chunk(const uint4* __restrict__ records, uint64_t rb, uint64_t re,
                                                     WaveStage<K>* __restrict__ st, int lane,
                                                     uint32_t* keys, uint32_t* cnt, uint32_t* ctxs, uint32_t* bcw,
                                                     uint32_t S, uint32_t* n_fill, uint32_t* overflow, uint32_t sub = 0,
                                                     uint32_t skip = 0, uint32_t quota = 0xFFFFFFFFu)
{
    // sub != 0: one of 2^(sub >> 8) sub-passes over a fine bucket too rich for one table -- only the k-mers whose
    // selector (a mix of the key words, 8 bits) equals sub & 0xFF are counted in this one
    // skip / quota: the wave's share of an item is a range of INSTANCES (k_count deals them out evenly); of the staged
    // piece it counts instances [skip, skip + quota) and returns how many that were
    const uint32_t sel_mask = (1u << (sub >> 8)) - 1u, sel = sub & 0xFFu;
    const uint32_t total = wave_stage_piece<K>(records, rb, re, st, lane);
    bool ok = true;
    uint32_t n_claimed = 0;
    const uint32_t total_u = __builtin_amdgcn_readfirstlane(total);     // scalar loop control
    const uint32_t first_u = __builtin_amdgcn_readfirstlane(min(skip, total_u));
    const uint32_t end_u = first_u + min(quota, total_u - first_u);
    for (uint32_t t0 = first_u; t0 < end_u; t0 += 64) {
        const uint32_t t = t0 + lane;
        Probe A = make_probe<K>(fetch_instance<K>(st, min(t, end_u - 1u)), S, t < end_u);
        // (every key word goes into the selector: the k-mers of a hot bucket share their minimizer, often at the
        // same offset, i.e. whole words of the key)
        if (SUB) A.active = A.active && ((((A.k0 * 0x9E3779B1u) ^ (A.k1 * 0x85EBCA77u) ^ (A.k2 * 0xC2B2AE3Du) ^ (A.k3 * 0x27D4EB2Fu)) >> 24) & sel_mask) == sel;
#ifdef DFK_ABLATE_INSERT        // timing experiment only: keep the extraction alive, skip the table
        if ((A.k0 ^ A.k1 ^ A.ctx) == 0x12345u) ++n_claimed;
#else
        ok = table_insert<KTraits<K>::KW, NBC, LDS_TABLE>(keys, cnt, ctxs, bcw, S, A, n_claimed) && ok;
#endif
    }
    n_claimed = wave_sum(n_claimed);
    if (lane == 0 && n_claimed) atomicAdd(n_fill, n_claimed);
    if (!ok) atomicOr(overflow, 1u);
    wave_sync();
    return end_u - first_u;
}

// ---- two consecutive k-mers of a record per lane: an EXPERIMENT (round 4), built with -DDFK_PAIRS, parity green, NOT the
// default -- measured on MI355X (tools/pairs_ab.sh, tools/pmc_sq.sh): k_count 963 ms per step as it is,
// 982 ms with the pair extraction and two inserts one after the other (-DDFK_PAIRS -DDFK_PAIRS_SERIAL), 993 ms with the two
// keys probed side by side (table_insert2).  The extraction block does shrink (127 vector instructions for two k-mers against
// 2 x 111), but a k_count wave is parked on s_waitcnt / barriers for 57 % of its cycles and issues during 31 % (SQ_WAIT_ANY,
// SQ_ACTIVE_INST_ANY): instructions are not what it is short of, and the side-by-side loop adds more of them (+11 % vector,
// +23 % scalar: both keys' bookkeeping in every iteration) than the shorter chain of LDS round trips gives back.
// Rebuilding a k-mer from the 2-bit stream is two fifths of the insert path's vector instructions (fetch ~12, extraction
// ~99 of ~267 per 64 instances), and nearly half of THAT is the 128-bit group reversal that turns the stream into KMer's
// big-endian order.  The k-mer that follows in the same record needs none of it: its forward value is the predecessor's
// shifted by one base with the next base appended -- KMer::toSuccessor, kmers/KMer.h:189-201, which is how the reference's own
// Kmerizer::map walks a read (BuildReadQGraph48.cc:148-165) -- and its reverse complement is the next window of the
// complemented stream.  So a lane takes instances (2j, 2j+1) of a record: one fetch, one full extraction, one rolled one,
// two inserts.  A record of nk k-mers takes ceil(nk/2) lane slots; with nk odd the last slot's second half is idle (one
// insert slot in 2 nk: 2 % at the 12.8 k-mers a record holds on average).
struct PairRegs { uint32_t hdr, tag, q; u128 X, F; };

// Stage records as wave_stage_piece does, indexing PAIR SLOTS instead of instances: returns the number of slots staged.
template <int K>
__device__ __forceinline__ uint32_t wave_stage_pairs(const uint4* __restrict__ records, uint64_t rb, uint64_t re,
                                                     WaveStage<K>* __restrict__ st, int lane)
{
    static_assert(COUNT_CHUNK == 32, "one uint4 per lane");
    const uint64_t hidx = 2 * rb + lane;
    uint4 v{0, 0, 0, 0};
    if (hidx < 2 * re) v = records[hidx];
    reinterpret_cast<uint4*>(st->rec)[lane] = v;
    st->msk[lane] = 0;
    wave_sync();
    const uint32_t nk = lane < COUNT_CHUNK ? (st->rec[8 * lane] & 63u) : 0u;
    const uint32_t np = (nk + 1u) >> 1;
    const uint32_t incl = wave_incl_scan(np, lane);
    const uint32_t start = incl - np;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (lane < COUNT_CHUNK) st->starts[lane] = start;
    if (np) atomicOr(&st->msk[start >> 5], 1u << (start & 31u));
    wave_sync();
    const uint32_t c0 = __popc(tld(&st->msk[lane]));
    st->pc[lane] = wave_incl_scan(c0, lane) - c0;
    wave_sync();
    return total;
}

// first k-mer of pair slot t: the full extraction (as make_probe), keeping the window and the forward value for the second
template <int K>
__device__ __forceinline__ Probe pair_first(const WaveStage<K>* __restrict__ st, uint32_t t, uint32_t S, bool active, PairRegs* keep)
{
    const uint32_t w = t >> 5;
    const uint32_t bits = tld(&st->msk[w]) & (0xFFFFFFFFu >> (31u - (t & 31u)));
    const uint32_t r = st->pc[w] + __popc(bits) - 1u;
    const uint32_t q = 2u * (t - st->starts[r]);
    const uint32_t* rec = st->rec + 8 * r;
    const uint32_t* p = rec + 2 + (q >> 4);
    const uint32_t hdr = rec[0], p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4];
    const uint32_t nk = hdr & 63u;
    // bits [2q, 2q + 128) of the payload: pred, K bases, succ, and the base after it (K + 3 bases: 2K + 6 <= 126 bits)
    const uint32_t sh = (2u * q) & 31u;
    const uint32_t x0 = alignbit(p1, p0, sh), x1 = alignbit(p2, p1, sh), x2 = alignbit(p3, p2, sh), x3 = alignbit(p4, p3, sh);
    const u128 X{(uint64_t)x0 | ((uint64_t)x1 << 32), (uint64_t)x2 | ((uint64_t)x3 << 32)};
    const uint32_t pred = x0 & 3u;
    const u128 ks = shr128(X, 2);
    constexpr int SB = 2 * K + 2;
    const uint32_t succ = (uint32_t)(SB >= 64 ? (X.hi >> (SB - 64)) : (X.lo >> SB)) & 3u;
    const u128 m = KTraits<K>::mask();
    const u128 R{~ks.lo & m.lo, ~ks.hi & m.hi};
    const u128 rv{rev2_64(ks.hi), rev2_64(ks.lo)};
    const u128 F = shr128(rv, 128 - KTraits<K>::BITS);                  // (the masked-off bases above the k-mer fall out at the bottom)
    const bool rev = lt128(R, F);
    uint32_t ctx = 0;
    if (q > 0 || (hdr & 64u)) ctx |= 0x10u << pred;
    if (q + 1 < nk || (hdr & 128u)) ctx |= 1u << succ;
    if (rev) ctx = ctx_rc(ctx);
    *keep = PairRegs{hdr, rec[1], q, X, F};
    return probe_begin(rev ? R : F, ctx, (int32_t)rec[1], S, active);
}

// second k-mer of the slot: KMer::toSuccessor on the forward value, the next window of the complemented stream
template <int K>
__device__ __forceinline__ Probe pair_second(const PairRegs& k, uint32_t S, bool active)
{
    const uint32_t nk = k.hdr & 63u, q = k.q + 1u;
    const u128 m = KTraits<K>::mask();
    constexpr int SB = 2 * K + 2;
    const uint32_t base_in = (uint32_t)(SB >= 64 ? (k.X.hi >> (SB - 64)) : (k.X.lo >> SB)) & 3u;       // the first k-mer's successor
    const uint32_t succ = (uint32_t)(SB + 2 >= 64 ? (k.X.hi >> (SB + 2 - 64)) : (k.X.lo >> (SB + 2))) & 3u;
    const uint32_t pred = (uint32_t)(k.X.lo >> 2) & 3u;                  // the first k-mer's first base
    const u128 ks = shr128(k.X, 4);
    const u128 R{~ks.lo & m.lo, ~ks.hi & m.hi};
    u128 F = shl128(k.F, 2); F.lo |= base_in; F.lo &= m.lo; F.hi &= m.hi;
    const bool rev = lt128(R, F);
    uint32_t ctx = 0x10u << pred;                                        // (q >= 1: there is a predecessor in the record)
    if (q + 1 < nk || (k.hdr & 128u)) ctx |= 1u << succ;
    if (rev) ctx = ctx_rc(ctx);
    return probe_begin(rev ? R : F, ctx, (int32_t)k.tag, S, active && q < nk);
}

template <int K, int NBC, bool SUB>
__device__ __forceinline__ void wave_count_chunk_pairs(const uint4* __restrict__ records, uint64_t rb, uint64_t re,
                                                       WaveStage<K>* __restrict__ st, int lane,
                                                       uint32_t* keys, uint32_t* cnt, uint32_t* ctxs, uint32_t* bcw,
                                                       uint32_t S, uint32_t* n_fill, uint32_t* overflow, uint32_t sub)
{
    const uint32_t sel_mask = (1u << (sub >> 8)) - 1u, sel = sub & 0xFFu;
    auto selected = [&](const Probe& A) { return ((((A.k0 * 0x9E3779B1u) ^ (A.k1 * 0x85EBCA77u) ^ (A.k2 * 0xC2B2AE3Du) ^ (A.k3 * 0x27D4EB2Fu)) >> 24) & sel_mask) == sel; };
    const uint32_t total = wave_stage_pairs<K>(records, rb, re, st, lane);
    bool ok = true;
    uint32_t n_claimed = 0;
    const uint32_t end_u = __builtin_amdgcn_readfirstlane(total);       // scalar loop control
    for (uint32_t t0 = 0; t0 < end_u; t0 += 64) {
        const uint32_t t = t0 + lane;
        PairRegs keep;
        Probe A = pair_first<K>(st, min(t, end_u - 1u), S, t < end_u, &keep);
        if (SUB) A.active = A.active && selected(A);
        Probe B = pair_second<K>(keep, S, t < end_u);
        if (SUB) B.active = B.active && selected(B);
#ifdef DFK_PAIRS_SERIAL
        ok = table_insert<KTraits<K>::KW, NBC, true>(keys, cnt, ctxs, bcw, S, A, n_claimed) && ok;
        ok = table_insert<KTraits<K>::KW, NBC, true>(keys, cnt, ctxs, bcw, S, B, n_claimed) && ok;
#else
        ok = table_insert2<KTraits<K>::KW, NBC>(keys, cnt, ctxs, bcw, S, A, B, n_claimed) && ok;
#endif
    }
    n_claimed = wave_sum(n_claimed);
    if (lane == 0 && n_claimed) atomicAdd(n_fill, n_claimed);
    if (!ok) atomicOr(overflow, 1u);
    wave_sync();
}

constexpr int BIG_TICKET_CHUNKS = 4;                 // chunks a wave takes per ticket

// item that owns position x of the concatenated chunk / slot space (pre[] ascending, pre[0] = 0, pre[n] = total)
__device__ __forceinline__ uint32_t big_find(const uint64_t* __restrict__ pre, uint32_t n, uint64_t x)
{
    uint32_t lo = 0, hi = n;                         // pre[lo] <= x < pre[hi]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}

// ---- a fine bucket too rich for one LDS table: second-level partition by k-mer hash
// All its instances are written out once as records of ONE k-mer each (same 32-byte format: nk = 1, the k-mer's own
// predecessor / successor flags), grouped by `sel` = the top log2p bits of a second hash of the canonical k-mer: every
// instance of a k-mer lands in the same sub-bucket, a sub-bucket holds ~700 distinct k-mers, and the sub-buckets are
// then counted by the ordinary k_count as the buckets of a small pass of their own.  Work linear in the bucket's
// instances -- counting it as 2^p sub-passes over the same records extracts every instance 2^p times.
struct HotItem { uint32_t b0, b1; uint32_t sub_base; uint32_t log2p; };

__device__ __forceinline__ uint32_t sub_bucket_hash(const Probe& A)
{
    uint32_t h = (A.k0 * 0x85EBCA77u) ^ __builtin_rotateleft32(A.k1 * 0xC2B2AE3Du, 7) ^ __builtin_rotateleft32(A.k2 * 0x27D4EB2Fu, 19) ^ (A.k3 * 0x9E3779B1u);
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 13;
    return h;
}

// one instance written out as a record of its own (the 32-byte format: nk = 1, its own predecessor / successor flags)
template <int K>
__device__ __forceinline__ void hot_write_record(const InstRegs& in, uint32_t sub, uint64_t dst, uint4* __restrict__ out)
{
    const uint32_t q = in.q, sh = (2u * q) & 31u, rnk = in.hdr & 63u;
    const bool hp = q > 0 || (in.hdr & 64u), hs = q + 1 < rnk || (in.hdr & 128u);
    uint32_t o[4] = {alignbit(in.p1, in.p0, sh), alignbit(in.p2, in.p1, sh), alignbit(in.p3, in.p2, sh), alignbit(in.p4, in.p3, sh)};
    const uint32_t nbits = 2u * (1u + (uint32_t)K + (hs ? 1u : 0u));      // <= 124
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t lo = 32u * w;
        if (nbits <= lo) o[w] = 0;
        else if (nbits < lo + 32u) o[w] &= (1u << (nbits - lo)) - 1u;
    }
    if (!hp) o[0] &= ~3u;
    out[2 * dst] = uint4{rec_header(1u, hp, hs, sub), in.tag, o[0], o[1]};
    out[2 * dst + 1] = uint4{o[2], o[3], 0u, 0u};
}

// The two passes of the second-level partition (WRITE = false: count the instances of every sub-bucket; WRITE = true:
// write every instance out as a record of its own, grouped by sub-bucket), with the counters of the hot bucket being
// read kept in LDS.  A workgroup takes HOT_BLOCK consecutive chunks -- almost always of one hot bucket: a bucket with
// 10^6 instances is 2400 chunks -- and counts their instances per sub-bucket with LDS atomics.  The counting pass then
// adds the non-zero counters to the global ones: 2^p global atomics per 50 000 instances instead of one per instance.
// The writing pass takes room for each sub-bucket's instances of this block with one global atomic per sub-bucket,
// leaves the positions in the LDS counters, and extracts the instances a second time to write them there (LDS atomic
// per instance): the records of one sub-bucket from one block are consecutive.  (With a global atomic per instance
// both passes ran at the rate of scattered HBM atomics: 1.3 s + 1.2 s of a repeat-rich human-scale step, against 0.24 s + 0.3 s.)
// Buckets with more than 2^HOT_LDS_LOG2 sub-buckets (a minimizer owning 10^7 distinct k-mers) keep the global atomics.
constexpr int HOT_BLOCK = 128;                       // chunks per workgroup ticket
constexpr uint32_t HOT_LDS_LOG2 = 14;                // sub-bucket counters a workgroup keeps in LDS (64 KB)
template <int K, int NWAVES, bool WRITE>
__global__ void __launch_bounds__(NWAVES * 64)
k_hot_pass(const uint4* __restrict__ records, const HotItem* __restrict__ items, const uint64_t* __restrict__ rec_base,
           const uint64_t* __restrict__ chunk_pre, uint32_t n_items, unsigned long long* __restrict__ ticket,
           unsigned long long* __restrict__ sub_acc,          // !WRITE: per sub-bucket records << 32 | instances
           unsigned long long* __restrict__ sub_cur,          // WRITE: per sub-bucket append cursor, starts at its first record
           const uint64_t* __restrict__ sub_first,            // WRITE: first record of every sub-bucket (the pass's bucket bases)
           uint64_t n_out, uint4* __restrict__ out)
{
    constexpr int NT = NWAVES * 64;
    __shared__ WaveStage<K> stages[NWAVES];
    __shared__ uint32_t hist[1u << HOT_LDS_LOG2];
    __shared__ unsigned long long blk_first;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    WaveStage<K>* st = &stages[wave];
    const uint64_t n_chunks = chunk_pre[n_items];
    for (uint32_t guard = 0; guard < 0x7FFFFFFFu; ++guard) {
        __syncthreads();                                               // (blk_first of the round before has been read)
        if (tid == 0) blk_first = atomicAdd(ticket, (unsigned long long)HOT_BLOCK);
        __syncthreads();
        const uint64_t first = uniform64((uint64_t)blk_first);
        if (first >= n_chunks) break;
        const uint64_t last = min(first + (uint64_t)HOT_BLOCK, n_chunks);
        uint64_t t = first;
        while (t < last) {                                             // one segment per hot bucket touched (uniform)
            const uint32_t it = big_find(chunk_pre, n_items, t);
            const HotItem I = items[it];
            const uint64_t c0 = chunk_pre[it], seg_end = min(last, chunk_pre[it + 1]);
            const uint64_t rb0 = rec_base[I.b0], re = rec_base[I.b1];
            const bool in_lds = I.log2p <= HOT_LDS_LOG2;
            const uint32_t n_sub = 1u << I.log2p;
            const uint64_t item_first = WRITE ? sub_first[I.sub_base] : 0ull;
            // a sweep over the segment's chunks; `mode` 0: count into LDS, 1: global atomics per instance (and write),
            // 2: write at the positions the LDS counters hold
            auto sweep = [&](const int mode) {
                for (uint64_t c = t + wave; c < seg_end; c += NWAVES) {
                    const uint64_t rb = rb0 + (c - c0) * COUNT_CHUNK;
                    const uint32_t total = wave_stage_piece<K>(records, rb, re, st, lane);
                    for (uint32_t i0 = 0; i0 < total; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        if (i < total) {
                            const InstRegs in = fetch_instance<K>(st, i);
                            const Probe A = make_probe<K>(in, 2u, true);
                            const uint32_t sub = I.log2p ? sub_bucket_hash(A) >> (32u - I.log2p) : 0u;
                            if (mode == 0) atomicAdd(&hist[sub], 1u);
                            else if (mode == 1) {
                                if (!WRITE) atomicAdd(&sub_acc[I.sub_base + sub], (1ull << 32) | 1ull);
                                else {
                                    const uint64_t dst = atomicAdd(&sub_cur[I.sub_base + sub], 1ull);
                                    if (dst < n_out) hot_write_record<K>(in, I.sub_base + sub, dst, out);
                                }
                            } else {
                                const uint64_t dst = item_first + atomicAdd(&hist[sub], 1u);
                                if (dst < n_out) hot_write_record<K>(in, I.sub_base + sub, dst, out);
                            }
                        }
                    }
                    wave_sync();
                }
            };
            if (!in_lds) sweep(1);
            else {
                for (uint32_t i = tid; i < n_sub; i += NT) hist[i] = 0;
                __syncthreads();
                sweep(0);
                __syncthreads();
                for (uint32_t i = tid; i < n_sub; i += NT) {
                    const uint32_t h = hist[i];
                    if (!WRITE) { if (h) atomicAdd(&sub_acc[I.sub_base + i], ((unsigned long long)h << 32) | h); }
                    else hist[i] = h ? (uint32_t)(atomicAdd(&sub_cur[I.sub_base + i], (unsigned long long)h) - item_first) : 0u;
                }
                __syncthreads();
                if (WRITE) { sweep(2); __syncthreads(); }
            }
            t = seg_end;
        }
    }
}

template <bool USE_BC>
__device__ __forceinline__ bool bc_pass(uint32_t v, uint32_t min_bc)
{
    if (!USE_BC || min_bc == 0) return true;
    if (min_bc == 1) return v != 0;
    return (v & BCW_MULTI) != 0;                                     // >= MIN_BC distinct barcodes > 0, or an ignored (-1) one
}

// Look a canonical k-mer up in a finished table (no concurrent inserts).  Returns its slot or ~0.
// One batch of reads per probe (count word and key words together), selects instead of branches.
template <int KW>
__device__ __forceinline__ uint32_t table_find(const uint32_t* keys, const uint32_t* cnt, uint32_t S, u128 c)
{
    const uint32_t k0 = (uint32_t)c.lo, k1 = (uint32_t)(c.lo >> 32), k2 = (uint32_t)c.hi, k3 = (uint32_t)(c.hi >> 32);
    const uint32_t h = key_hash(c);
    const uint32_t fp = (h >> 24) | 1u;
    const uint32_t step = ((h >> 11) | 1u) & (S - 1);
    uint32_t slot = h & (S - 1), found = ~0u, open = 1u;
    for (uint32_t p = 0; p <= COUNT_MAX_PROBE + 1 && open; ++p) {
        const uint32_t cw = tld(&cnt[slot]);
        const uint32_t r0 = tld(&keys[slot]), r1 = tld(&keys[S + slot]), r2 = tld(&keys[2 * S + slot]);
        const uint32_t r3 = KW == 4 ? tld(&keys[3 * S + slot]) : k3;
        const uint32_t match = (cw != 0u) & ((cw >> 24) == fp) & (r0 == k0) & (r1 == k1) & (r2 == k2) & (r3 == k3);
        found = match ? slot : found;
        open = (cw != 0u) & (match ^ 1u);
        slot = (slot + step) & (S - 1);
    }
    return found;
}

// canonical 2K-bit value of a k-mer given as a 2K-bit big-endian value
template <int K>
__device__ __forceinline__ u128 canon_value(u128 F)
{
    const u128 m = KTraits<K>::mask();
    u128 nf{~F.lo & m.lo, ~F.hi & m.hi};
    u128 top = shl128(nf, 128 - KTraits<K>::BITS);
    u128 R{rev2_64(top.hi), rev2_64(top.lo)};
    return lt128(R, F) ? R : F;
}

// ctl words (LDS)
enum { CTL_ITEM = 0, CTL_OVF = 1, CTL_FILL = 2, CTL_CHUNK = 3, CTL_USED = 4, CTL_DISTINCT = 5, CTL_NTASK = 6, CTL_BOUNDARY = 7,
       CTL_RB_LO = 8, CTL_RB_HI = 9, CTL_RE_LO = 10, CTL_RE_HI = 11, CTL_NSOLID = 12,
       CTL_OUT_LO = 13, CTL_OUT_HI = 14, CTL_NEXT_LO = 15, CTL_NEXT_HI = 16,
       CTL_B0 = 17, CTL_B1 = 18,                     // fine buckets [b0, b1) of the item being counted
       CTL_SP = 19, CTL_SUB = 20,                    // ... and its sub-pass word (wave_count_chunk)
       CTL_STACK = 24, CTL_STACK_CAP = 16,           // bucket ranges waiting to be counted by this workgroup (b0, b1, sub)
       CTL_N = CTL_STACK + 3 * CTL_STACK_CAP };

// Finish a counted LDS table: decide solidity, clean up adjacencies, emit.  (The HBM-table fallback does
// the same three steps as separate grid-wide launches: k_big_flags / k_big_resolve / k_big_emit.)
//
// recomputeAdjacencies (ReadPather.h:329-364) keeps a context bit only if the neighbouring k-mer is
// solid.  A neighbour observed next to this k-mer in a read almost always shares its minimizer bucket
// and therefore THIS table, which holds every instance of every k-mer it contains.  So:
//   pass 1  every slot: solid?  the barcode word becomes {FLAG_SOLID | unresolved bits}; each set context
//           bit of a solid slot becomes a task (slot, bit) in an LDS queue
//   pass 2  tasks, densely over the threads: neighbour found in this table -> keep the bit iff that slot
//           is solid (final answer); not found -> the neighbour lives in another item: bit stays, marked
//           unresolved for the small HBM pass afterwards (k_adjacency) -- about one bit in ten
//   pass 3  emit solid slots into the workgroup's chunk of the dictionary part (WgOut) and the spectrum
// `sync` is __syncthreads for the whole workgroup.
template <int K, int NBC, uint32_t ADJ_TASKS>
__device__ __forceinline__ uint32_t table_finish(uint32_t* keys, uint32_t* cnt, uint32_t* ctxs, uint32_t* bcw,
                                                 uint32_t S, const CountParams& cp, uint4* __restrict__ seg_out,
                                                 uint32_t* ctl, unsigned long long* part_cursor, unsigned int* seg_overflow,
                                                 uint32_t* hist_lds, unsigned long long* __restrict__ hist_global,
                                                 uint32_t* tasks, uint32_t* n_tasks, uint32_t* n_boundary,
                                                 uint16_t* solid_list, uint32_t* n_solid, int tid, int nthreads
#ifdef DFK_PHASE_TIMES
                                                 , PhaseClock& phase
#endif
                                                 )
{
    constexpr int KW = KTraits<K>::KW;
    const int lane = tid & 63;
    uint32_t n_occ = 0;
    const u128 m = KTraits<K>::mask();
    // one queued look-up: is the neighbour of `slot` across context bit `bit` in this table, and solid?
    auto resolve = [&](uint32_t task) {
        const uint32_t slot = task & 0x0FFFFFFFu, bit = task >> 28;
        u128 F{(uint64_t)tld(&keys[slot]) | ((uint64_t)tld(&keys[S + slot]) << 32),
               (uint64_t)tld(&keys[2 * S + slot]) | (KW == 4 ? ((uint64_t)tld(&keys[3 * S + slot]) << 32) : 0ull)};
        u128 v;
        if (bit < 4) { v = shl128(F, 2); v.lo &= m.lo; v.hi &= m.hi; v.lo |= bit; }         // kmer[1:] + base
        else {                                                                                 // base + kmer[:-1]
            v = shr128(F, 2);
            constexpr int TOP = KTraits<K>::BITS - 2;
            if (TOP >= 64) v.hi |= (uint64_t)(bit - 4) << (TOP - 64); else v.lo |= (uint64_t)(bit - 4) << TOP;
        }
        const uint32_t f = table_find<KW>(keys, cnt, S, canon_value<K>(v));
        if (f == ~0u) atomicOr(&bcw[slot], 1u << bit);                                         // lives in another item
        else if (!(tld(&bcw[f]) & FLAG_SOLID)) atomicAnd(&ctxs[slot], ~(1u << bit));           // here, and not solid
    };
    auto sync = [&]() { __syncthreads(); };
    // ---- pass 1: every thread takes S / nthreads slots; their state words are read first (independent loads).
    // Solid slots are few (one slot in twenty), so the work on them is done on dense lanes: the wave reserves room in
    // the list of solid slots once (wave scan of the lanes' needs, one atomic by the last lane) and writes its slots
    // there; then the lanes take the wave's own entries back, one each, and queue their adjacency look-ups the same
    // way (scan of the popcounts, one atomic, a loop over the set bits).  As it was -- per solid slot a returning
    // atomic on ONE LDS word and an 8-step bit loop at three active lanes -- this pass took 17 % of all wave cycles.
    {
        constexpr uint32_t PER = 4;                                     // slots per thread and round
        for (uint32_t base = 0; base < S; base += PER * nthreads) {
            uint32_t c[PER], b[PER];
#pragma unroll
            for (uint32_t k = 0; k < PER; ++k) {
                const uint32_t slot = base + k * nthreads + tid;
                c[k] = slot < S ? tld(&cnt[slot]) : 0u;
                b[k] = (NBC > 0 && slot < S) ? tld(&bcw[slot]) : 0u;
            }
            uint32_t solid_m = 0, need = 0;
#pragma unroll
            for (uint32_t k = 0; k < PER; ++k) {
                const uint32_t count = c[k] & CNT_MASK;                 // saturated at 2^24-1 by the insert (ReadPather.h:128-129)
                const bool occ = c[k] != 0u;
                const bool solid = occ && count >= cp.min_freq && bc_pass<(NBC > 0)>(b[k], cp.min_bc);
                n_occ += occ;
                if (solid) { solid_m |= 1u << k; ++need; }
            }
            const uint32_t incl = wave_incl_scan(need, lane);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t base_s = 0;
            if (lane == 63 && total) base_s = atomicAdd(n_solid, total);
            base_s = (uint32_t)__builtin_amdgcn_readlane((int)base_s, 63);
            uint32_t ps = base_s + incl - need;
#pragma unroll
            for (uint32_t k = 0; k < PER; ++k) {
                const uint32_t slot = base + k * nthreads + tid;
                if (c[k] == 0u) continue;
                const bool solid = (solid_m >> k) & 1u;
                if (solid) solid_list[ps++] = (uint16_t)slot;
                tst(&bcw[slot], solid ? FLAG_SOLID : 0u);
            }
            if (cp.do_adj && total) {
                wave_sync();                                            // the wave's own list entries, written by other lanes
                for (uint32_t i0 = 0; i0 < total; i0 += 64) {
                    const uint32_t i = i0 + (uint32_t)lane;
                    const bool act = i < total;
                    const uint32_t slot = act ? (uint32_t)solid_list[base_s + i] : 0u;
                    uint32_t x = act ? (tld(&ctxs[slot]) & 0xFFu) : 0u;
                    if (cp.keep_pre && act) tst(&ctxs[slot], x | (x << 8));
                    const uint32_t nb = __popc(x);
                    const uint32_t incl2 = wave_incl_scan(nb, lane);
                    const uint32_t tot2 = (uint32_t)__builtin_amdgcn_readlane((int)incl2, 63);
                    uint32_t base_t = 0;
                    if (lane == 63 && tot2) base_t = atomicAdd(n_tasks, tot2);
                    base_t = (uint32_t)__builtin_amdgcn_readlane((int)base_t, 63);
                    uint32_t pt = base_t + incl2 - nb;
                    if (pt + nb <= ADJ_TASKS)
                        while (x) { const uint32_t bit = (uint32_t)__builtin_ctz(x); tasks[pt++] = slot | (bit << 28); x &= x - 1u; }
                }
            }
        }
    }
    PH(4);
    sync();
    PH(5);
    // ---- room for this item's solid k-mers: the rest of the workgroup's chunk, and a fresh one if that is not enough
    constexpr uint32_t OUT_CHUNK = 4 * ADJ_TASKS;                      // = 2 S entries: more than a table can emit
    if (tid == 0) {
        const uint32_t ns = tld(n_solid), used = tld(&ctl[CTL_USED]);
        if (used + ns > OUT_CHUNK) {
            const unsigned long long nx = atomicAdd(part_cursor, (unsigned long long)OUT_CHUNK);
            tst(&ctl[CTL_NEXT_LO], (uint32_t)nx); tst(&ctl[CTL_NEXT_HI], (uint32_t)(nx >> 32));
        }
    }
    // ---- pass 2
#ifdef DFK_ABLATE_PASS2
    if (false) {
#else
    if (cp.do_adj) {
#endif
        const uint32_t total = __builtin_amdgcn_readfirstlane(tld(n_tasks));
        if (total <= ADJ_TASKS) {
            for (uint32_t t = tid; t < total; t += nthreads) resolve(tasks[t]);
            PH(6);
            sync();
            PH(7);
        } else {
            // more look-ups than the queue holds (a table full of solid k-mers): redo the queueing in slot
            // ranges that cannot overflow it (<= 8 look-ups per slot)
            for (uint32_t lo = 0; lo < S; lo += ADJ_TASKS / 8) {
                __syncthreads();
                if (tid == 0) tst(n_tasks, 0u);
                __syncthreads();
                for (uint32_t slot = lo + tid; slot < min(S, lo + ADJ_TASKS / 8); slot += nthreads) {
                    if (!(tld(&cnt[slot]) && (tld(&bcw[slot]) & FLAG_SOLID))) continue;
                    const uint32_t ctx = tld(&ctxs[slot]) & 0xFFu;
                    uint32_t pos = atomicAdd(n_tasks, __popc(ctx));
                    for (uint32_t bit = 0; bit < 8; ++bit) if (ctx & (1u << bit)) tasks[pos++] = slot | (bit << 28);
                }
                __syncthreads();
                const uint32_t n = __builtin_amdgcn_readfirstlane(tld(n_tasks));
                for (uint32_t t = tid; t < n; t += nthreads) resolve(tasks[t]);
            }
            sync();
        }
    }
    else sync();                                                      // (the chunk claimed above must be visible)
    // ---- pass 3
    uint32_t boundary = 0;
    auto emit = [&](uint32_t slot, uint32_t c, uint32_t flags, unsigned long long idx) {
        const uint32_t count = c & CNT_MASK;
        const uint32_t cw = tld(&ctxs[slot]);
        const uint32_t pending = flags & 0xFFu & cw;                    // a bit cleared locally needs no further look-up
        boundary += pending != 0;
        if (idx < cp.seg_cap) {
            u128 v{(uint64_t)tld(&keys[slot]) | ((uint64_t)tld(&keys[S + slot]) << 32),
                   (uint64_t)tld(&keys[2 * S + slot]) | (KW == 4 ? ((uint64_t)tld(&keys[3 * S + slot]) << 32) : 0ull)};
            u128 kw = shl128(v, 128 - KTraits<K>::BITS);               // left-align: KMer<K> storage
            seg_out[2 * idx] = uint4{(uint32_t)kw.hi, (uint32_t)(kw.hi >> 32), (uint32_t)kw.lo, (uint32_t)(kw.lo >> 32)};
            // pad (word 3) carries the unresolved bits (and the original context for tests) until k_adjacency
            seg_out[2 * idx + 1] = uint4{0xFFFFFFFFu, count | ((cw & 0xFFu) << 24), 0xFFFFFFFFu, pending | (cw & 0xFF00u)};
        } else atomicOr(seg_overflow, 1u);
        if (count < (uint32_t)COUNT_HIST_BINS) atomicAdd(&hist_lds[count], 1u);
        else atomicAdd(&hist_global[count], 1ull);
    };
    {
        // the solid slots were listed by pass 1, so the emit runs on dense lanes, into the room thread 0 took for
        // them above; then the state words of the whole table are cleared with wide stores (key words are
        // rewritten on claim).  cnt, ctxs and the barcode words are contiguous.
        const uint32_t ns = __builtin_amdgcn_readfirstlane(tld(n_solid));
        const uint32_t used = __builtin_amdgcn_readfirstlane(tld(&ctl[CTL_USED]));
        const unsigned long long cur = uniform64(tld(&ctl[CTL_OUT_LO]), tld(&ctl[CTL_OUT_HI]));
        const unsigned long long nxt = uniform64(tld(&ctl[CTL_NEXT_LO]), tld(&ctl[CTL_NEXT_HI]));
#ifdef DFK_ABLATE_EMIT
        for (uint32_t i = tid; i < 0 * ns; i += nthreads) {
#else
        for (uint32_t i = tid; i < ns; i += nthreads) {
#endif
            const uint32_t slot = solid_list[i];
            const uint32_t pos = used + i;
            emit(slot, tld(&cnt[slot]), tld(&bcw[slot]), pos < OUT_CHUNK ? cur + pos : nxt + (pos - OUT_CHUNK));
        }
        PH(8);
        __syncthreads();
        PH(9);
        if (tid == 0) {
            if (used + ns > OUT_CHUNK) { tst(&ctl[CTL_OUT_LO], (uint32_t)nxt); tst(&ctl[CTL_OUT_HI], (uint32_t)(nxt >> 32)); tst(&ctl[CTL_USED], used + ns - OUT_CHUNK); }
            else tst(&ctl[CTL_USED], used + ns);
            tst(n_solid, 0u);
        }
        uint4* z = reinterpret_cast<uint4*>(cnt);
        for (uint32_t i = tid; i < (3 + (NBC > 1 ? NBC - 1 : 0)) * S / 4; i += nthreads) z[i] = uint4{0, 0, 0, 0};
    }
    boundary = wave_sum(boundary);
    if (lane == 0 && boundary) atomicAdd(n_boundary, boundary);
    PH(10);
    return n_occ;
}


template <int K, int LOG2S, int NWAVES, int NBC, bool SUB>    // SUB: the launch over sub-passes of single buckets (item_sub given)
__global__ void __launch_bounds__(NWAVES * 64)
k_count(const uint4* __restrict__ records, const ItemRange* __restrict__ items, const uint64_t* __restrict__ rec_base,
        CountParams cp, CountGlobals* __restrict__ g, uint4* __restrict__ out, WgOut* __restrict__ wg_out,
        unsigned long long* __restrict__ hist_global, ItemRange* __restrict__ overflow_items,
        const uint32_t* __restrict__ item_sub,        // optional: per item, its sub-pass word (single buckets counted in several sub-passes)
        unsigned int* __restrict__ resident)          // bumped once by every workgroup as it starts (k_gate waits on it)
{
    constexpr uint32_t S = 1u << LOG2S;
    constexpr int KW = KTraits<K>::KW;
    constexpr int NT = NWAVES * 64;
    extern __shared__ __align__(16) uint32_t smem[];
    uint32_t* keys = smem;                          // [KW][S]
    uint32_t* cnt = keys + KW * S;                  // [S]
    uint32_t* ctxs = cnt + S;                       // [S]
    uint32_t* bcw = ctxs + S;                       // [max(1, NBC)][S]
    constexpr uint32_t XW = NBC > 1 ? NBC - 1 : 0;  // barcode words beyond the first
    uint32_t* hist = bcw + (1 + XW) * S;            // [COUNT_HIST_BINS]
    uint32_t* ctl = hist + COUNT_HIST_BINS;         // [CTL_N]
    uint32_t* tasks = ctl + CTL_N;                  // [S] neighbour look-up queue
    uint16_t* solid_list = reinterpret_cast<uint16_t*>(tasks + S / 2); // [S] solid slots of the item being finished  (tasks: [S/2])
    WaveStage<K>* stages = reinterpret_cast<WaveStage<K>*>(tasks + S / 2 + S / 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    WaveStage<K>* st = stages + wave;
    uint4* seg_out = out;                                              // the part's reservation, shared by all workgroups

    // This workgroup has its LDS: say so.  The next range's sweep (thousands of short-lived 4-KB blocks on the second stream)
    // is held back by k_gate until every persistent workgroup of this launch has said it -- a sweep block placed in the
    // middle of a CU's LDS before the CU's second workgroup arrives leaves no 71-KB hole, and that workgroup then never
    // starts: the launch runs at half rate to its end ("the cliff").
    if (tid == 0 && resident) __hip_atomic_fetch_add(resident, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = tid; i < COUNT_HIST_BINS; i += NT) hist[i] = 0;
    for (uint32_t i = tid; i < (3 + XW) * S; i += NT) cnt[i] = 0;      // count, context and barcode words; key words are written on claim
    // The item loop is software-pipelined: while the workgroup counts item i, thread 0 already holds the
    // ticket and the record range of item i+1 in registers (a returning global atomic plus a dependent
    // load are ~4 us of latency that every wave would otherwise wait for behind a barrier).
    // An item that overflows its table is cut in two by bucket index on the spot: thread 0 keeps a small stack
    // of bucket ranges in LDS (the halves, and the ticket it already holds) and the workgroup works that off
    // before it takes another ticket.  Only single buckets that overflow go back to the host (HBM tables).
    if (tid == 0) {
        const WgOut w = wg_out[blockIdx.x];
        ctl[CTL_OUT_LO] = (uint32_t)w.chunk; ctl[CTL_OUT_HI] = (uint32_t)(w.chunk >> 32); ctl[CTL_USED] = w.used;
        ctl[CTL_NEXT_LO] = 0; ctl[CTL_NEXT_HI] = 0; ctl[CTL_DISTINCT] = 0; ctl[CTL_BOUNDARY] = 0;
        ctl[CTL_OVF] = 0; ctl[CTL_FILL] = 0; ctl[CTL_CHUNK] = 0; ctl[CTL_NTASK] = 0; ctl[CTL_NSOLID] = 0; ctl[CTL_SP] = 0;
        const uint32_t it0 = atomicAdd(&g->next_item, 1u);
        uint64_t b0 = 0, e0 = 0; ItemRange r0{0, 0}; uint32_t s0 = 0;
        if (it0 < cp.n_items) { r0 = items[it0]; b0 = rec_base[r0.b0]; e0 = rec_base[r0.b1]; if (SUB) s0 = item_sub[it0]; }
        ctl[CTL_ITEM] = it0; ctl[CTL_RB_LO] = (uint32_t)b0; ctl[CTL_RB_HI] = (uint32_t)(b0 >> 32);
        ctl[CTL_RE_LO] = (uint32_t)e0; ctl[CTL_RE_HI] = (uint32_t)(e0 >> 32);
        ctl[CTL_B0] = r0.b0; ctl[CTL_B1] = r0.b1; ctl[CTL_SUB] = s0;
    }
#ifdef DFK_PHASE_TIMES
    PhaseClock phase; phase.start();
#endif
    uint32_t items_done = 0;
    for (;;) {
        PH(0);
        __syncthreads();                                               // table empty, counters reset, item published
        PH(1);
        const uint32_t item = __builtin_amdgcn_readfirstlane(ctl[CTL_ITEM]);
        if (item >= cp.n_items) break;
        ++items_done;
        const uint64_t rb = uniform64(ctl[CTL_RB_LO], ctl[CTL_RB_HI]);
        const uint64_t re = uniform64(ctl[CTL_RE_LO], ctl[CTL_RE_HI]);
        const uint32_t sub = __builtin_amdgcn_readfirstlane(ctl[CTL_SUB]);
        // (thread 0 keeps the prefetched ticket in registers: storing it to LDS here would make its wave wait
        // for the loads at the top of every item -- measured, +1 % on the whole kernel)
        uint32_t nx = 0xFFFFFFFFu; uint64_t nb = 0, ne = 0; ItemRange nr{0, 0}; uint32_t nsub = 0;
        if (tid == 0 && ctl[CTL_SP] == 0) {                            // (ranges on the stack come first)
            nx = atomicAdd(&g->next_item, 1u);
            if (nx < cp.n_items) { nr = items[nx]; nb = rec_base[nr.b0]; ne = rec_base[nr.b1]; if (SUB) nsub = item_sub[nx]; }
        }
        // Waves pull chunks of the item from an LDS ticket.  All loop control is made scalar
        // (readfirstlane) so the compiler emits uniform branches, and the trip count is bounded.
        // The item's records are dealt to the waves in equal consecutive shares, worked off in pieces of COUNT_CHUNK.
        // (They used to be pulled as 32-record chunks from an LDS ticket: an item is 7-9 such chunks, so one wave of the
        // eight often took a second chunk while seven waited at the barrier below -- 22 % of all wave cycles.)
        const uint32_t n_rec = (uint32_t)(re - rb);
        const uint32_t share = (n_rec + NWAVES - 1) / NWAVES;
        const uint32_t wv = __builtin_amdgcn_readfirstlane((uint32_t)wave);       // (scalar loop control)
        const uint32_t w_lo = min(n_rec, wv * share), w_hi = min(n_rec, w_lo + share);
        if (!SUB && cp.single) {
            // one-k-mer records (k_hot_pass wrote them): a lane takes a record straight from HBM -- staged 32 at a
            // time they would fill half the lanes of a batch
            for (uint32_t at = w_lo; at < w_hi; at += 64) {
                if (__builtin_amdgcn_readfirstlane(tld(&ctl[CTL_OVF]))) break;
                const uint32_t r = at + (uint32_t)lane;
                uint4 a{0, 0, 0, 0}, b{0, 0, 0, 0};
                if (r < w_hi) { a = records[2 * (rb + r)]; b = records[2 * (rb + r) + 1]; }
                const InstRegs in{a.x, a.y, a.z, a.w, b.x, b.y, b.z, 0u};
                const Probe A = make_probe<K>(in, S, (a.x & 63u) != 0u);
                uint32_t claimed = 0;
                const bool ok = table_insert<KW, NBC, true>(keys, cnt, ctxs, bcw, S, A, claimed);
                claimed = wave_sum(claimed);
                if (lane == 0 && claimed) atomicAdd(&ctl[CTL_FILL], claimed);
                if (!ok) atomicOr(&ctl[CTL_OVF], 1u);
                if (lane == 0 && tld(&ctl[CTL_FILL]) > (S / 4) * 3) tst(&ctl[CTL_OVF], 1u);
            }
        } else
        for (uint32_t at = w_lo; at < w_hi; at += COUNT_CHUNK) {
            if (__builtin_amdgcn_readfirstlane(tld(&ctl[CTL_OVF]))) break;
#ifndef DFK_PAIRS
            wave_count_chunk<K, NBC, true, SUB>(records, rb + at, rb + w_hi, st, lane, keys, cnt, ctxs, bcw, S,
                                             &ctl[CTL_FILL], &ctl[CTL_OVF], sub);
#else           // experiment, measured and not kept (below): two consecutive k-mers per lane, -DDFK_PAIRS_SERIAL or probed side by side
            wave_count_chunk_pairs<K, NBC, SUB>(records, rb + at, rb + w_hi, st, lane, keys, cnt, ctxs, bcw, S,
                                                &ctl[CTL_FILL], &ctl[CTL_OVF], sub);
#endif
            if (lane == 0 && tld(&ctl[CTL_FILL]) > (S / 4) * 3) tst(&ctl[CTL_OVF], 1u);   // stop when 3/4 full
        }
        PH(2);
        __syncthreads();
        PH(3);
        if (__builtin_amdgcn_readfirstlane(ctl[CTL_OVF])) {
            if (tid == 0) {
                const uint32_t b0 = ctl[CTL_B0], b1 = ctl[CTL_B1];
                uint32_t sp = ctl[CTL_SP];
                if (b1 - b0 > 1 && sp + 3 <= (uint32_t)CTL_STACK_CAP) {
                    if (nx < cp.n_items) { ctl[CTL_STACK + 3 * sp] = nr.b0; ctl[CTL_STACK + 3 * sp + 1] = nr.b1; ctl[CTL_STACK + 3 * sp + 2] = nsub; ++sp; nx = 0xFFFFFFFFu; }
                    const uint32_t mid = b0 + (b1 - b0) / 2;
                    ctl[CTL_STACK + 3 * sp] = mid; ctl[CTL_STACK + 3 * sp + 1] = b1; ctl[CTL_STACK + 3 * sp + 2] = 0u; ++sp;
                    ctl[CTL_STACK + 3 * sp] = b0; ctl[CTL_STACK + 3 * sp + 1] = mid; ctl[CTL_STACK + 3 * sp + 2] = 0u; ++sp;
                    ctl[CTL_SP] = sp;
                    atomicAdd(&g->n_split, 1u);
                } else overflow_items[atomicAdd(&g->n_overflow, 1u)] = ItemRange{b0, SUB ? (0x80000000u | ctl[CTL_SUB]) : b1};   // (a sub-pass reports which one it was)
            }
            for (uint32_t i = tid; i < (3 + XW) * S; i += NT) cnt[i] = 0;   // abandon the table
            __syncthreads();                                           // everyone has read CTL_OVF before it is reset
        } else {
#ifdef DFK_ABLATE_FINISH        // timing experiment only: no solidity/adjacency/emit passes
            for (uint32_t i = tid; i < (3 + XW) * S; i += NT) cnt[i] = 0;
            uint32_t occ = 0;
            if (false)
#else
            uint32_t occ =
#endif
            table_finish<K, NBC, S / 2>(keys, cnt, ctxs, bcw, S, cp, seg_out, ctl, &g->part_cursor,
                                                         &g->solid_overflow, hist, hist_global, tasks, &ctl[CTL_NTASK],
                                                         &ctl[CTL_BOUNDARY], solid_list, &ctl[CTL_NSOLID], tid, NT
#ifdef DFK_PHASE_TIMES
                                                         , phase
#endif
                                                         );
            occ = wave_sum(occ);
            if (lane == 0 && occ) atomicAdd(&ctl[CTL_DISTINCT], occ);
        }
        if (tid == 0) {                                                // publish the next item (stack first), reset the per-item words
            const uint32_t sp = ctl[CTL_SP];
            uint32_t n0 = nr.b0, n1 = nr.b1;
            if (sp) {
                n0 = ctl[CTL_STACK + 3 * (sp - 1)]; n1 = ctl[CTL_STACK + 3 * (sp - 1) + 1]; nsub = ctl[CTL_STACK + 3 * (sp - 1) + 2];
                ctl[CTL_SP] = sp - 1;
                nx = 0; nb = rec_base[n0]; ne = rec_base[n1];
            }
            ctl[CTL_ITEM] = nx; ctl[CTL_RB_LO] = (uint32_t)nb; ctl[CTL_RB_HI] = (uint32_t)(nb >> 32);
            ctl[CTL_RE_LO] = (uint32_t)ne; ctl[CTL_RE_HI] = (uint32_t)(ne >> 32);
            ctl[CTL_B0] = n0; ctl[CTL_B1] = n1; ctl[CTL_SUB] = nsub;
            ctl[CTL_OVF] = 0; ctl[CTL_FILL] = 0; ctl[CTL_CHUNK] = 0; ctl[CTL_NTASK] = 0;
        }
    }
    __syncthreads();
#ifdef DFK_PHASE_TIMES
    PH(11); phase.flush(lane);
#endif
    for (int i = tid; i < COUNT_HIST_BINS; i += NT) if (hist[i]) atomicAdd(&hist_global[i], (unsigned long long)hist[i]);
    if (tid == 0) {
        wg_out[blockIdx.x] = WgOut{(unsigned long long)ctl[CTL_OUT_LO] | ((unsigned long long)ctl[CTL_OUT_HI] << 32), ctl[CTL_USED], 0u};
        if (ctl[CTL_DISTINCT]) atomicAdd(&g->n_distinct, (unsigned long long)ctl[CTL_DISTINCT]);
        if (ctl[CTL_BOUNDARY]) atomicAdd(&g->n_boundary, (unsigned long long)ctl[CTL_BOUNDARY]);
        if (!items_done) atomicAdd(&g->n_wg_idle, 1u);
    }
}

// One wave, no LDS, on the sweep's stream in front of the sweep: returns when the counter has reached `target` (every
// persistent workgroup of the k_count launched just before has its LDS) or after `max_ticks` of the 100 MHz wall clock,
// whichever comes first -- a bounded wait, then the sweep goes ahead regardless.
__global__ void __launch_bounds__(64)
k_gate(const unsigned int* __restrict__ ctr, unsigned int target, unsigned long long max_ticks, unsigned int* __restrict__ timed_out)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        if (wall_clock64() - t0 > max_ticks) { __hip_atomic_fetch_add(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(64);
    }
}

template <int K, int LOG2S, int NWAVES, int NBC>
constexpr size_t count_lds_bytes()
{
    return sizeof(uint32_t) * ((size_t)(KTraits<K>::KW + 3 + (NBC > 1 ? NBC - 1 : 0)) * (1u << LOG2S) + COUNT_HIST_BINS + CTL_N + (1u << LOG2S) / 2 + (1u << LOG2S) / 2) + sizeof(WaveStage<K>) * NWAVES;
}

// Fallback for a fine bucket that cannot fit an LDS table: its table lives in HBM (tab = [KW+3][S] words,
// zeroed by the host, load <= 0.5) and the WHOLE GRID works on it -- a single minimizer can own a sizeable
// share of a real genome's k-mers (low-complexity and repeat-derived m-mers), so one workgroup per such
// bucket would serialise the run.  Four launches over all fallback items of a pass together:
//   k_big_insert   waves pull chunks of 32 records through a global ticket and insert with the same code
//                  as k_count (atomics at agent scope on the HBM table)
//   k_big_flags    per slot: solid or not
//   k_big_resolve  per solid slot and context bit: neighbour in the same table?  (as table_finish pass 2)
//   k_big_emit     solid slots to the fallback's output buffer through a global cursor, spectrum, counters
struct BigItem { uint32_t b0, b1; uint64_t tab_off; uint32_t log2s; uint32_t pad; };
template <int K, int NWAVES, int NBC>
__global__ void __launch_bounds__(NWAVES * 64)
k_big_insert(const uint4* __restrict__ records, const BigItem* __restrict__ items, const uint64_t* __restrict__ rec_base,
             const uint64_t* __restrict__ chunk_pre, uint32_t n_items, uint32_t* __restrict__ tab_pool,
             unsigned long long* __restrict__ ticket, uint32_t* __restrict__ failed)
{
    constexpr int KW = KTraits<K>::KW;
    __shared__ WaveStage<K> stages[NWAVES];
    __shared__ uint32_t fill, ovf;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { fill = 0; ovf = 0; }
    __syncthreads();
    const uint64_t n_chunks = chunk_pre[n_items];
    for (uint32_t guard = 0; guard < 0x7FFFFFFFu; ++guard) {                 // every wave leaves when the tickets run out
        unsigned long long t0 = 0;
        if (lane == 0) t0 = atomicAdd(ticket, (unsigned long long)BIG_TICKET_CHUNKS);
        const uint64_t first = uniform64((uint64_t)t0);
        if (first >= n_chunks) break;
        for (int k = 0; k < BIG_TICKET_CHUNKS; ++k) {
            const uint64_t t = first + k;
            if (t >= n_chunks) break;
            const uint32_t it = big_find(chunk_pre, n_items, t);
            const BigItem I = items[it];
            const uint32_t S = 1u << I.log2s;
            uint32_t* keys = tab_pool + I.tab_off;
            uint32_t* cnt = keys + (size_t)KW * S;
            const uint64_t rb = rec_base[I.b0] + (t - chunk_pre[it]) * COUNT_CHUNK;
            wave_count_chunk<K, NBC, false>(records, rb, rec_base[I.b1], &stages[wave], lane, keys, cnt, cnt + S, cnt + 3 * (size_t)S, S, &fill, &ovf);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && ovf) atomicOr(failed, 1u);
}

// words of an HBM table: [KW][S] keys, [S] state (fingerprint << 24, see table_insert), [S] contexts, [S] counts, then the barcode words
template <int K> struct BigView {
    uint32_t* keys; uint32_t* cnt; uint32_t* ctxs; uint32_t* wide; uint32_t* bcw; uint32_t S, slot;
    __device__ __forceinline__ uint32_t count() const { const uint32_t c = tld(&wide[slot]); return c < CNT_MASK ? c : CNT_MASK; }   // KDef::setCount saturates at 2^24-1
};
template <int K>
__device__ __forceinline__ BigView<K> big_view(const BigItem* __restrict__ items, const uint64_t* __restrict__ slot_pre, uint32_t n_items,
                                               uint32_t* __restrict__ tab_pool, uint64_t x)
{
    const uint32_t it = big_find(slot_pre, n_items, x);
    const BigItem I = items[it];
    const uint32_t S = 1u << I.log2s;
    uint32_t* keys = tab_pool + I.tab_off;
    uint32_t* cnt = keys + (size_t)KTraits<K>::KW * S;
    return BigView<K>{keys, cnt, cnt + S, cnt + 2 * (size_t)S, cnt + 3 * (size_t)S, S, (uint32_t)(x - slot_pre[it])};
}

template <int K, int NBC>
__global__ void __launch_bounds__(256)
k_big_flags(const BigItem* __restrict__ items, const uint64_t* __restrict__ slot_pre, uint32_t n_items, uint32_t* __restrict__ tab_pool,
            CountParams cp, CountGlobals* __restrict__ g)
{
    const uint64_t total = slot_pre[n_items];
    uint32_t occ = 0;
    for (uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x; x < total; x += (uint64_t)gridDim.x * 256) {
        const BigView<K> v = big_view<K>(items, slot_pre, n_items, tab_pool, x);
        const uint32_t c = tld(&v.cnt[v.slot]);
        if (!c) continue;
        ++occ;
        const bool solid = v.count() >= cp.min_freq && bc_pass<(NBC > 0)>(NBC > 0 ? tld(&v.bcw[v.slot]) : 0u, cp.min_bc);
        if (solid && cp.do_adj && cp.keep_pre) { const uint32_t ctx = tld(&v.ctxs[v.slot]) & 0xFFu; tst(&v.ctxs[v.slot], ctx | (ctx << 8)); }
        tst(&v.bcw[v.slot], solid ? FLAG_SOLID : 0u);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) occ += __shfl_down(occ, d, 64);
    if ((threadIdx.x & 63) == 0 && occ) atomicAdd(&g->n_distinct, (unsigned long long)occ);
}

template <int K>
__global__ void __launch_bounds__(256)
k_big_resolve(const BigItem* __restrict__ items, const uint64_t* __restrict__ slot_pre, uint32_t n_items, uint32_t* __restrict__ tab_pool)
{
    constexpr int KW = KTraits<K>::KW;
    const uint64_t total = slot_pre[n_items];
    const u128 m = KTraits<K>::mask();
    for (uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x; x < total; x += (uint64_t)gridDim.x * 256) {
        const BigView<K> v = big_view<K>(items, slot_pre, n_items, tab_pool, x);
        if (!tld(&v.cnt[v.slot]) || !(tld(&v.bcw[v.slot]) & FLAG_SOLID)) continue;
        const uint32_t S = v.S, slot = v.slot;
        const u128 F{(uint64_t)tld(&v.keys[slot]) | ((uint64_t)tld(&v.keys[S + slot]) << 32),
                     (uint64_t)tld(&v.keys[2 * (size_t)S + slot]) | (KW == 4 ? ((uint64_t)tld(&v.keys[3 * (size_t)S + slot]) << 32) : 0ull)};
        for (uint32_t ctx = tld(&v.ctxs[slot]) & 0xFFu; ctx; ctx &= ctx - 1) {
            const uint32_t bit = (uint32_t)__ffs(ctx) - 1u;
            u128 nb;
            if (bit < 4) { nb = shl128(F, 2); nb.lo &= m.lo; nb.hi &= m.hi; nb.lo |= bit; }
            else {
                nb = shr128(F, 2);
                constexpr int TOP = KTraits<K>::BITS - 2;
                if (TOP >= 64) nb.hi |= (uint64_t)(bit - 4) << (TOP - 64); else nb.lo |= (uint64_t)(bit - 4) << TOP;
            }
            const uint32_t f = table_find<KW>(v.keys, v.cnt, S, canon_value<K>(nb));
            if (f == ~0u) atomicOr(&v.bcw[slot], 1u << bit);                                      // lives in another item
            else if (!(tld(&v.bcw[f]) & FLAG_SOLID)) atomicAnd(&v.ctxs[slot], ~(1u << bit));      // here, and not solid
        }
    }
}

template <int K>
__global__ void __launch_bounds__(256)
k_big_emit(const BigItem* __restrict__ items, const uint64_t* __restrict__ slot_pre, uint32_t n_items, uint32_t* __restrict__ tab_pool,
           CountParams cp, CountGlobals* __restrict__ g, uint4* __restrict__ out, unsigned long long* __restrict__ hist_global)
{
    constexpr int KW = KTraits<K>::KW;
    const uint64_t total = slot_pre[n_items];
    const int lane = threadIdx.x & 63;
    uint32_t boundary = 0;
    const uint64_t rounds = (total + (uint64_t)gridDim.x * 256 - 1) / ((uint64_t)gridDim.x * 256);
    for (uint64_t r = 0; r < rounds; ++r) {                                    // whole waves stay together for the ballot
        const uint64_t x = (r * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        bool solid = false; BigView<K> v{}; uint32_t c = 0, flags = 0;
        if (x < total) {
            v = big_view<K>(items, slot_pre, n_items, tab_pool, x);
            c = tld(&v.cnt[v.slot]);
            flags = c ? tld(&v.bcw[v.slot]) : 0u;
            solid = (flags & FLAG_SOLID) != 0;
        }
        const unsigned long long mk = __ballot(solid);
        if (!mk) continue;
        unsigned long long wbase = 0;
        if (lane == 0) wbase = atomicAdd(&g->big_cursor, (unsigned long long)__popcll(mk));
        wbase = uniform64((uint64_t)wbase);
        if (!solid) continue;
        const uint32_t S = v.S, slot = v.slot, count = v.count();
        const uint32_t cw = tld(&v.ctxs[slot]);
        const uint32_t pending = flags & 0xFFu & cw;
        boundary += pending != 0;
        const unsigned long long idx = wbase + __popcll(mk & ((1ull << lane) - 1ull));
        if (idx < cp.seg_cap) {
            const u128 kv{(uint64_t)tld(&v.keys[slot]) | ((uint64_t)tld(&v.keys[S + slot]) << 32),
                          (uint64_t)tld(&v.keys[2 * (size_t)S + slot]) | (KW == 4 ? ((uint64_t)tld(&v.keys[3 * (size_t)S + slot]) << 32) : 0ull)};
            const u128 kw = shl128(kv, 128 - KTraits<K>::BITS);
            out[2 * idx] = uint4{(uint32_t)kw.hi, (uint32_t)(kw.hi >> 32), (uint32_t)kw.lo, (uint32_t)(kw.lo >> 32)};
            out[2 * idx + 1] = uint4{0xFFFFFFFFu, count | ((cw & 0xFFu) << 24), 0xFFFFFFFFu, pending | (cw & 0xFF00u)};
        } else atomicOr(&g->solid_overflow, 1u);
        atomicAdd(&hist_global[count], 1ull);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) boundary += __shfl_down(boundary, d, 64);
    if (lane == 0 && boundary) atomicAdd(&g->n_boundary, (unsigned long long)boundary);
}

// After a pass: entries at the tail of the handed-out region move into the holes the workgroups left at the end
// of their last chunks (src and dst are entry indices, built by the host from the WgOut records).
__global__ void __launch_bounds__(256)
k_fill_holes(uint4* __restrict__ out, const uint64_t* __restrict__ src, const uint64_t* __restrict__ dst, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[2 * dst[i]] = out[2 * src[i]];
    out[2 * dst[i] + 1] = out[2 * src[i] + 1];
}

// Order-independent digest of a run of dictionary entries: digest[0] += sum of h(entry), digest[1] ^= xor of
// h'(entry), h = a 64-bit mix of all 32 bytes.  For checking at sizes where the entries cannot be compared one
// by one (the same dictionary through different pass geometries, single GPU against sharded): digests of
// disjoint sets add / xor.  tests/util.py computes the same value in numpy from the oracle's entries.
__device__ __forceinline__ uint64_t digest_mix(uint64_t x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void __launch_bounds__(256)
k_digest(const uint4* __restrict__ entries, uint64_t n, unsigned long long* __restrict__ digest)
{
    uint64_t sum = 0, xr = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 a = entries[2 * i], b = entries[2 * i + 1];
        const uint64_t w0 = a.x | ((uint64_t)a.y << 32), w1 = a.z | ((uint64_t)a.w << 32);
        const uint64_t w2 = b.x | ((uint64_t)b.y << 32), w3 = b.z | ((uint64_t)b.w << 32);
        const uint64_t h = digest_mix(w0 ^ digest_mix(w1 ^ digest_mix(w2 ^ digest_mix(w3 + 0x9E3779B97F4A7C15ull))));
        sum += h; xr ^= digest_mix(h + 0xD1B54A32D192ED03ull);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { sum += __shfl_down(sum, d, 64); xr ^= __shfl_down(xr, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&digest[0], sum); atomicXor(&digest[1], xr); }
}

// ============================================================================ a6: adjacency clean-up
// KmerDict::recomputeAdjacencies (ReadPather.h:329-364): a context bit survives only if the
// neighbouring canonical k-mer is itself solid.  The solid set is put in an HBM open-addressing
// set (16-byte keys, empty = w0 all ones, which no canonical k-mer can have), then every solid
// entry probes its <= 8 neighbours.
struct SetSlot { uint64_t w0, w1; };

__device__ __forceinline__ uint64_t set_hash(uint64_t w0, uint64_t w1)
{
    uint64_t h = w0 * 0x9E3779B97F4A7C15ull ^ (w1 + 0x7F4A7C159E3779B9ull) * 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 29;
    return h;
}

__global__ void __launch_bounds__(256)
k_set_insert(const uint4* __restrict__ entries, uint64_t n, SetSlot* __restrict__ set, uint64_t mask)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!(entries[2 * i + 1].w & 0xFFu)) return;                     // only k-mers with unresolved context bits can be asked for
    uint4 a = entries[2 * i];
    uint64_t w0 = (uint64_t)a.x | ((uint64_t)a.y << 32), w1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
    uint64_t s = set_hash(w0, w1) & mask;
    for (;;) {
        unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&set[s].w0), ~0ull, (unsigned long long)w0);
        if (old == ~0ull) { set[s].w1 = w1; return; }
        s = (s + 1) & mask;
    }
}

__device__ __forceinline__ bool set_has(const SetSlot* __restrict__ set, uint64_t mask, uint64_t w0, uint64_t w1)
{
    uint64_t s = set_hash(w0, w1) & mask;
    for (;;) {
        SetSlot v = set[s];
        if (v.w0 == ~0ull) return false;
        if (v.w0 == w0 && v.w1 == w1) return true;
        s = (s + 1) & mask;
    }
}

// The entries of a part that still have unresolved context bits (pad byte 0), as a dense list of indices: built
// once per pass while the next pass is counted, so that the set build and the look-ups after the last pass touch
// those entries only (about one in eight) instead of streaming the whole dictionary twice.
// ctl[0] = entries listed so far, ctl[1] = set if the list would overflow `cap`.
constexpr int BLIST_PER_THREAD = 8;
__global__ void __launch_bounds__(256)
k_boundary_list(const uint4* __restrict__ entries, uint64_t n, uint32_t* __restrict__ list, uint64_t cap,
                unsigned long long* __restrict__ ctl)
{
    __shared__ uint32_t found[256 * BLIST_PER_THREAD];
    __shared__ uint32_t n_found;
    __shared__ unsigned long long at;
    const int lane = threadIdx.x & 63;
    const uint64_t per_block = 256ull * BLIST_PER_THREAD;
    for (uint64_t b0 = (uint64_t)blockIdx.x * per_block; b0 < n; b0 += (uint64_t)gridDim.x * per_block) {
        if (threadIdx.x == 0) n_found = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < BLIST_PER_THREAD; ++j) {
            const uint64_t i = b0 + 256ull * j + threadIdx.x;
            const bool hit = i < n && (entries[2 * i + 1].w & 0xFFu) != 0u;
            const unsigned long long mk = __ballot(hit);
            uint32_t w = 0;
            if (lane == 0 && mk) w = atomicAdd(&n_found, (uint32_t)__popcll(mk));
            w = __builtin_amdgcn_readfirstlane(w);
            if (hit) found[w + __popcll(mk & ((1ull << lane) - 1ull))] = (uint32_t)i;
        }
        __syncthreads();
        const uint32_t m = n_found;
        if (threadIdx.x == 0) at = m ? atomicAdd(&ctl[0], (unsigned long long)m) : 0ull;
        __syncthreads();
        const unsigned long long base = at;
        if (base + m > cap) { if (threadIdx.x == 0) atomicOr(&ctl[1], 1ull); }
        else for (uint32_t t = threadIdx.x; t < m; t += 256) list[base + t] = found[t];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
k_set_insert_list(const uint4* __restrict__ entries, const uint32_t* __restrict__ list, uint64_t n, SetSlot* __restrict__ set, uint64_t mask)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint4 a = entries[2 * (uint64_t)list[t]];
    const uint64_t w0 = (uint64_t)a.x | ((uint64_t)a.y << 32), w1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
    uint64_t s = set_hash(w0, w1) & mask;
    for (uint64_t step = 0; step <= mask; ++step) {                      // (the host sizes the set at load <= 0.6; a full one must still let every wave end)
        unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&set[s].w0), ~0ull, (unsigned long long)w0);
        if (old == ~0ull) { set[s].w1 = w1; return; }
        s = (s + 1) & mask;
    }
}

// canonical left-aligned (w0,w1) of a k-mer given as a 2K-bit big-endian value
template <int K>
__device__ __forceinline__ void canon_words(u128 F, uint64_t* w0, uint64_t* w1)
{
    // reverse complement of a big-endian value: group-reverse the complement
    const u128 m = KTraits<K>::mask();
    u128 nf{~F.lo & m.lo, ~F.hi & m.hi};
    u128 top = shl128(nf, 128 - KTraits<K>::BITS);
    u128 R{rev2_64(top.hi), rev2_64(top.lo)};
    u128 c = lt128(R, F) ? R : F;
    u128 kw = shl128(c, 128 - KTraits<K>::BITS);
    *w0 = kw.hi; *w1 = kw.lo;
}

// Only the context bits k_count could not settle inside its table (pad byte 0 of the entry) are looked up;
// the set holds exactly the solid k-mers that have such bits (adjacency is mutual: if X's neighbour Y was
// counted in another item, then Y's neighbour X was too).  Returns the number of look-ups.
template <int K>
__device__ __forceinline__ uint32_t adjacency_entry(uint4* __restrict__ entries, uint64_t i, const SetSlot* __restrict__ set, uint64_t mask)
{
    const u128 m = KTraits<K>::mask();
    uint4 b = entries[2 * i + 1];
    const uint32_t pending = b.w & 0xFFu;
    if (!pending) return 0;
    uint4 a = entries[2 * i];
    u128 kw{(uint64_t)a.z | ((uint64_t)a.w << 32), (uint64_t)a.x | ((uint64_t)a.y << 32)};
    u128 F = shr128(kw, 128 - KTraits<K>::BITS);                  // 2K-bit big-endian value
    uint32_t ctx = b.y >> 24, probes = 0;
    for (uint32_t bit = 0; bit < 8; ++bit) if (pending & (1u << bit)) {
        u128 v;
        if (bit < 4) { v = shl128(F, 2); v.lo &= m.lo; v.hi &= m.hi; v.lo |= bit; }            // kmer.toSuccessor
        else {
            v = shr128(F, 2);                                                                     // kmer.toPredecessor
            constexpr int TOP = KTraits<K>::BITS - 2;
            if (TOP >= 64) v.hi |= (uint64_t)(bit - 4) << (TOP - 64); else v.lo |= (uint64_t)(bit - 4) << TOP;
        }
        uint64_t w0, w1; canon_words<K>(v, &w0, &w1);
        ++probes;
        if (!set_has(set, mask, w0, w1)) ctx &= ~(1u << bit);
    }
    b.y = (b.y & 0xFFFFFFu) | (ctx << 24);
    b.w = 0;
    entries[2 * i + 1] = b;
    return probes;
}

__device__ __forceinline__ void add_probes(unsigned long long probes, unsigned long long* __restrict__ n_probes)
{
    __shared__ unsigned long long sh[4];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) probes += __shfl_down(probes, d, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = probes;
    __syncthreads();
    if (threadIdx.x == 0 && (sh[0] + sh[1] + sh[2] + sh[3])) atomicAdd(n_probes, sh[0] + sh[1] + sh[2] + sh[3]);
}

template <int K>
__global__ void __launch_bounds__(256)
k_adjacency(uint4* __restrict__ entries, uint64_t n, const SetSlot* __restrict__ set, uint64_t mask,
            unsigned long long* __restrict__ n_probes)
{
    unsigned long long probes = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        probes += adjacency_entry<K>(entries, i, set, mask);
    add_probes(probes, n_probes);
}

// the same over a part's boundary list (k_boundary_list): every lane has an entry to settle
template <int K>
__global__ void __launch_bounds__(256)
k_adjacency_list(uint4* __restrict__ entries, const uint32_t* __restrict__ list, uint64_t n, const SetSlot* __restrict__ set, uint64_t mask,
                 unsigned long long* __restrict__ n_probes)
{
    unsigned long long probes = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x)
        probes += adjacency_entry<K>(entries, (uint64_t)list[t], set, mask);
    add_probes(probes, n_probes);
}

// Test support (DFK_F_KEEP_PRE_ADJ): the kmers.kvec view = entries with their original context byte
// (kept in pad byte 1 by k_count); clears the pad of both copies' source bits it consumed.
__global__ void __launch_bounds__(256)
k_make_pre(uint4* __restrict__ entries, uint4* __restrict__ pre, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 a = entries[2 * i], b = entries[2 * i + 1];
    pre[2 * i] = a;
    pre[2 * i + 1] = uint4{b.x, (b.y & 0xFFFFFFu) | (((b.w >> 8) & 0xFFu) << 24), b.z, 0u};
    b.w &= 0xFFu;
    entries[2 * i + 1] = b;
}

__global__ void __launch_bounds__(256)
k_clear_pad(uint4* __restrict__ entries, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* w = reinterpret_cast<uint32_t*>(entries + 2 * i + 1) + 3;
    if (*w) *w = 0;
}

// ============================================================================ multi-GPU adjacency exchange
// Fine bucket of an arbitrary k-mer given as a 2K-bit big-endian value (same function of the
// canonical m-mer set as k_partition computes while scanning a read).
template <int K>
__device__ __forceinline__ uint32_t kmer_bucket(u128 F, const PartParams& pp)
{
    const uint32_t M = pp.M;
    const uint32_t mmask = M == 16 ? 0xFFFFFFFFu : ((1u << (2 * M)) - 1u);
    const uint32_t rsh = 2 * (M - 1);
    uint32_t f = 0, rc = 0, mv = 0xFFFFFFFFu;
    for (int i = 0; i < K; ++i) {
        const int sh = 2 * (K - 1 - i);
        const uint32_t b = (uint32_t)(sh >= 64 ? (F.hi >> (sh - 64)) : (F.lo >> sh)) & 3u;
        f = ((f << 2) | b) & mmask;
        rc = (rc >> 2) | ((3u - b) << rsh);
        if (i + 1 >= (int)M) { uint32_t h = mmer_hash(f < rc ? f : rc); mv = h < mv ? h : mv; }
    }
    uint32_t bucket = (mv * 0x9E3779B1u) >> (32 - pp.log2_nb);
    return bucket;                       // owner rank = bucket & (world-1)
}

// Neighbour look-ups the local tables could not settle, grouped by the rank that owns each neighbour.
// Three dense steps (about one solid k-mer in ten has such a look-up, and finding the owner of a neighbour
// means sliding a minimizer window over it: done on sparse lanes that was most of a sharded run's adjacency
// time):
//   k_adj_list   entries -> list of (entry index << 3 | context bit)
//   k_adj_owner  list -> canonical neighbour key and owner rank of every item; items per owner
//   k_adj_place  keys and sources written grouped by owner
// Blocks work in rounds of ADJ_ROUND items and touch a global counter once per round and owner
// (same-address global atomics run at ~88 per microsecond chip-wide).
constexpr int ADJ_PER_THREAD = 8, ADJ_ROUND = 256 * ADJ_PER_THREAD;

// exclusive prefix of v over the 256 threads of a block; *total = block sum.  scratch: 8 words of LDS
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* scratch, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += scratch[w];
    *total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    __syncthreads();
    return off + incl - v;
}

// The same regroup without a global atomic: two levels of counting sort, all counters in LDS.
//   level 1  RG_PARTS partitions by the top bits of the bucket id.  A block owns a contiguous chunk of the input:
//            it counts its records per partition (k_rg_hist), a device scan of the [partition][block] counts gives
//            every block its own write position in every partition, and the block copies its records there
//            (k_rg_scatter; the positions are LDS cursors).
//   level 2  one workgroup per partition (k_rg_finish): counts the records and instances of its <= RG_FINE fine
//            buckets in LDS -- these are the pass's bucket counters, written out in the counting scan's format --
//            scans them and moves every record to its bucket's place inside the partition (again LDS cursors).
// 6 x 32 B of streaming traffic per record instead of two scattered global atomics and a scattered store.
constexpr uint32_t RG_LOG2_PARTS = 12, RG_PARTS = 1u << RG_LOG2_PARTS;
constexpr uint32_t RG_FINE = 4096;                 // fine buckets per partition at most (2^24 buckets per pass / RG_PARTS)

__global__ void __launch_bounds__(256)
k_rg_hist(const uint4* __restrict__ in, uint64_t n, uint64_t chunk, uint32_t local_mask, uint32_t part_shift, uint32_t n_parts,
          uint64_t* __restrict__ h1)               // [n_parts][gridDim.x] (+ one trailing zero, written by the host)
{
    __shared__ uint32_t hist[RG_PARTS];
    for (uint32_t i = threadIdx.x; i < n_parts; i += 256) hist[i] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256)
        atomicAdd(&hist[((in[2 * i].x >> 8) & local_mask) >> part_shift], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_parts; i += 256) h1[(uint64_t)i * gridDim.x + blockIdx.x] = hist[i];
}

__global__ void __launch_bounds__(256)
k_rg_scatter(const uint4* __restrict__ in, uint64_t n, uint64_t chunk, uint32_t local_mask, uint32_t part_shift, uint32_t n_parts,
             const uint64_t* __restrict__ h1off, uint4* __restrict__ out)
{
    __shared__ unsigned long long cur[RG_PARTS];
    for (uint32_t i = threadIdx.x; i < n_parts; i += 256) cur[i] = h1off[(uint64_t)i * gridDim.x + blockIdx.x];
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const uint4 a = in[2 * i], b = in[2 * i + 1];
        const unsigned long long dst = atomicAdd(&cur[((a.x >> 8) & local_mask) >> part_shift], 1ull);
        out[2 * dst] = a; out[2 * dst + 1] = b;
    }
}

__global__ void __launch_bounds__(256)
k_rg_finish(const uint4* __restrict__ in, uint64_t n, uint32_t local_mask, uint32_t part_shift, uint32_t n_parts, uint32_t n_blocks1,
            const uint64_t* __restrict__ h1off, unsigned long long* __restrict__ bucket_acc, uint4* __restrict__ out)
{
    __shared__ uint32_t rec[RG_FINE], inst[RG_FINE];
    __shared__ uint32_t scratch[8];
    const uint32_t fine = 1u << part_shift, fmask = fine - 1u;              // fine buckets per partition (<= RG_FINE)
    for (uint32_t p = blockIdx.x; p < n_parts; p += gridDim.x) {
        const uint64_t s = h1off[(uint64_t)p * n_blocks1], e = p + 1 < n_parts ? h1off[(uint64_t)(p + 1) * n_blocks1] : n;
        for (uint32_t i = threadIdx.x; i < fine; i += 256) { rec[i] = 0; inst[i] = 0; }
        __syncthreads();
        for (uint64_t i = s + threadIdx.x; i < e; i += 256) {
            const uint32_t h = in[2 * i].x, f = (h >> 8) & fmask;
            atomicAdd(&rec[f], 1u); atomicAdd(&inst[f], h & 63u);
        }
        __syncthreads();
        // the bucket counters of the pass (records << 32 | instances), then rec[] becomes each bucket's first index
        uint32_t carry = 0;
        for (uint32_t f0 = 0; f0 < fine; f0 += 256) {
            const uint32_t f = f0 + threadIdx.x;
            const uint32_t r = f < fine ? rec[f] : 0u;
            if (f < fine) bucket_acc[(uint64_t)p * fine + f] = ((unsigned long long)r << 32) | inst[f];
            uint32_t total;
            const uint32_t ex = block_excl_scan_256(r, scratch, &total);
            if (f < fine) rec[f] = carry + ex;
            carry += total;
        }
        __syncthreads();
        for (uint64_t i = s + threadIdx.x; i < e; i += 256) {
            const uint4 a = in[2 * i], b = in[2 * i + 1];
            const uint64_t dst = s + atomicAdd(&rec[(a.x >> 8) & fmask], 1u);
            out[2 * dst] = a; out[2 * dst + 1] = b;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
k_adj_list(const uint4* __restrict__ entries, uint64_t n, unsigned long long* __restrict__ n_items, uint64_t* __restrict__ list)
{
    __shared__ uint32_t scratch[8];
    __shared__ unsigned long long base;
    for (uint64_t i0 = (uint64_t)blockIdx.x * ADJ_ROUND; i0 < n; i0 += (uint64_t)gridDim.x * ADJ_ROUND) {
        uint32_t pend[ADJ_PER_THREAD], mine = 0;
#pragma unroll
        for (int k = 0; k < ADJ_PER_THREAD; ++k) {
            const uint64_t i = i0 + 256ull * k + threadIdx.x;
            pend[k] = i < n ? entries[2 * i + 1].w & 0xFFu : 0u;
            mine += __popc(pend[k]);
        }
        uint32_t total;
        uint32_t at = block_excl_scan_256(mine, scratch, &total);
        if (threadIdx.x == 0) base = total && list ? atomicAdd(n_items, (unsigned long long)total) : 0ull;
        if (!list) { if (threadIdx.x == 0 && total) atomicAdd(n_items, (unsigned long long)total); continue; }   // counting launch
        __syncthreads();
        const unsigned long long b = base;
#pragma unroll
        for (int k = 0; k < ADJ_PER_THREAD; ++k) {
            const uint64_t i = i0 + 256ull * k + threadIdx.x;
            for (uint32_t p = pend[k]; p; p &= p - 1) list[b + at++] = (i << 3) | (uint32_t)(__ffs(p) - 1);
        }
        __syncthreads();
    }
}

template <int K>
__global__ void __launch_bounds__(256)
k_adj_owner(const uint4* __restrict__ entries, const uint64_t* __restrict__ list, uint64_t n_items, PartParams pp,
            SetSlot* __restrict__ tmp_keys, uint8_t* __restrict__ owner, unsigned long long* __restrict__ per_owner)
{
    __shared__ unsigned int cnt[256];
    const uint32_t world = 1u << pp.log2_world;
    const u128 m = KTraits<K>::mask();
    if (threadIdx.x < world) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n_items; j += (uint64_t)gridDim.x * 256) {
        const uint64_t s = list[j];
        const uint32_t bit = (uint32_t)s & 7u;
        const uint4 a = entries[2 * (s >> 3)];
        const u128 kw{(uint64_t)a.z | ((uint64_t)a.w << 32), (uint64_t)a.x | ((uint64_t)a.y << 32)};
        const u128 F = shr128(kw, 128 - KTraits<K>::BITS);
        u128 v;
        if (bit < 4) { v = shl128(F, 2); v.lo &= m.lo; v.hi &= m.hi; v.lo |= bit; }                // kmer[1:] + base
        else {                                                                                      // base + kmer[:-1]
            v = shr128(F, 2);
            constexpr int TOP = KTraits<K>::BITS - 2;
            if (TOP >= 64) v.hi |= (uint64_t)(bit - 4) << (TOP - 64); else v.lo |= (uint64_t)(bit - 4) << TOP;
        }
        const uint32_t o = kmer_bucket<K>(v, pp) & (world - 1);
        uint64_t w0, w1; canon_words<K>(v, &w0, &w1);
        tmp_keys[j] = SetSlot{w0, w1};
        owner[j] = (uint8_t)o;
        atomicAdd(&cnt[o], 1u);
    }
    __syncthreads();
    if (threadIdx.x < world && cnt[threadIdx.x]) atomicAdd(&per_owner[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}

__global__ void __launch_bounds__(256)
k_adj_place(const uint64_t* __restrict__ list, const SetSlot* __restrict__ tmp_keys, const uint8_t* __restrict__ owner, uint64_t n_items,
            uint32_t world, const uint64_t* __restrict__ base, unsigned long long* __restrict__ fill,
            SetSlot* __restrict__ keys, uint64_t* __restrict__ src)
{
    __shared__ unsigned int cnt[256];
    __shared__ unsigned long long start[256];
    for (uint64_t j0 = (uint64_t)blockIdx.x * ADJ_ROUND; j0 < n_items; j0 += (uint64_t)gridDim.x * ADJ_ROUND) {
        if (threadIdx.x < world) cnt[threadIdx.x] = 0;
        __syncthreads();
        uint32_t own[ADJ_PER_THREAD];
#pragma unroll
        for (int k = 0; k < ADJ_PER_THREAD; ++k) {
            const uint64_t j = j0 + 256ull * k + threadIdx.x;
            own[k] = j < n_items ? owner[j] : 0xFFFFFFFFu;
            if (own[k] != 0xFFFFFFFFu) atomicAdd(&cnt[own[k]], 1u);
        }
        __syncthreads();
        if (threadIdx.x < world) {
            const unsigned int c = cnt[threadIdx.x];
            start[threadIdx.x] = base[threadIdx.x] + (c ? atomicAdd(&fill[threadIdx.x], (unsigned long long)c) : 0ull);
            cnt[threadIdx.x] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ADJ_PER_THREAD; ++k) {
            const uint64_t j = j0 + 256ull * k + threadIdx.x;
            if (own[k] == 0xFFFFFFFFu) continue;
            const uint64_t pos = start[own[k]] + atomicAdd(&cnt[own[k]], 1u);
            keys[pos] = tmp_keys[j];
            src[pos] = list[j];
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
k_adj_answer(const SetSlot* __restrict__ keys, uint64_t n, const SetSlot* __restrict__ set, uint64_t mask,
             uint8_t* __restrict__ present)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SetSlot k = keys[i];
    present[i] = set_has(set, mask, k.w0, k.w1) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
k_adj_apply(uint4* __restrict__ entries, const uint64_t* __restrict__ src, const uint8_t* __restrict__ present, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || present[i]) return;
    const uint64_t s = src[i];
    uint32_t* word = reinterpret_cast<uint32_t*>(entries + 2 * (s >> 3) + 1) + 1;    // count_ctx
    atomicAnd(word, ~(1u << (24 + (uint32_t)(s & 7))));
}

// ============================================================================ multi-GPU: regroup received records
// After the all-to-all a rank holds the records it owns from every peer, in arrival order.
// Items must be ranges of whole fine buckets, so scatter them once more by the fine bucket id
// kept in the header (pure 32-byte record moves).
template <bool WRITE>
__global__ void __launch_bounds__(256)
k_regroup(const uint4* __restrict__ in, uint64_t n, uint32_t local_mask, unsigned long long* __restrict__ bucket_acc,
          const uint64_t* __restrict__ bucket_base, uint32_t* __restrict__ bucket_cur, uint4* __restrict__ out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 a = in[2 * i];
    uint32_t b = (a.x >> 8) & local_mask;
    if (!WRITE) atomicAdd(&bucket_acc[b], (1ull << 32) | (a.x & 63u));
    else {
        uint64_t dst = bucket_base[b] + atomicAdd(&bucket_cur[b], 1u);
        out[2 * dst] = a; out[2 * dst + 1] = in[2 * i + 1];
    }
}


// Reads whose run summary overflowed (more than SUMMARY_RUNS runs), in ascending order: they go through the
// scanning scatter, lane per read, so the list must be dense.
__global__ void __launch_bounds__(256)
k_select_overflow(const uint4* __restrict__ summaries, uint64_t n_reads, uint32_t* __restrict__ list,
                  unsigned long long* __restrict__ n_list)
{
    // each block owns one contiguous slice of the reads: count its keepers, reserve the range with ONE global
    // atomic, then write them in order (one atomic per 256 reads on a single address cost 84 ms)
    __shared__ unsigned long long base;
    __shared__ uint32_t wcnt[4];
    const uint32_t* first = reinterpret_cast<const uint32_t*>(summaries);
    const uint64_t per = (n_reads + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n_reads ? lo + per : n_reads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t mine = 0;
    for (uint64_t r = lo + threadIdx.x; r < hi; r += blockDim.x) mine += (first[4 * r] & 15u) == SUMMARY_OVERFLOW;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d, 64);
    if (lane == 0) wcnt[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]; base = t ? atomicAdd(n_list, (unsigned long long)t) : 0ull; }
    __syncthreads();
    if (wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3] == 0) return;             // the usual case: nothing to list (block-uniform)
    unsigned long long at = base;
    for (uint64_t r0 = lo; r0 < hi; r0 += blockDim.x) {
        const uint64_t r = r0 + threadIdx.x;
        const bool keep = r < hi && (first[4 * r] & 15u) == SUMMARY_OVERFLOW;
        const unsigned long long m = __ballot(keep);
        __syncthreads();
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        uint32_t off = 0;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        if (keep) list[at + off + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)r;
        at += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    }
}

// ============================================================================ device-side bucket tables
// Everything that is O(number of fine buckets) stays on the GPU: a 30x human set has ~10^8 of them.

// exclusive scan of u64, 2048 elements per block; block totals go to `sums` (scanned by the next level)
constexpr int SCAN_THREADS = 256, SCAN_ITEMS = 8;
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_blocks(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t n, uint64_t* __restrict__ sums)
{
    __shared__ uint64_t wsum[SCAN_THREADS / 64];
    const uint64_t base = ((uint64_t)blockIdx.x * SCAN_THREADS + threadIdx.x) * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS], run = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = base + i < n ? in[base + i] : 0; run += v[i]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t incl = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint64_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    uint64_t ex = woff + incl - run;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
    if (threadIdx.x == SCAN_THREADS - 1 && sums) sums[blockIdx.x] = woff + incl;
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_add(uint64_t* __restrict__ out, uint64_t n, const uint64_t* __restrict__ sums)
{
    const uint64_t add = sums[blockIdx.x];
    const uint64_t base = ((uint64_t)blockIdx.x * SCAN_THREADS + threadIdx.x) * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) out[base + i] += add;
}

// after a scatter: every bucket received exactly the records the counting scan saw for it
__global__ void __launch_bounds__(256)
k_check_cursors(const unsigned long long* __restrict__ cur, const uint64_t* __restrict__ base, uint64_t nb, unsigned int* __restrict__ bad)
{
    unsigned int wrong = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b0 < nb; b0 += 4 * stride) {
        unsigned long long c[4]; uint64_t e[4];                        // eight loads in flight per lane
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint64_t b = b0 + k * stride; c[k] = b < nb ? cur[b] : 0ull; e[k] = b < nb ? base[b + 1] : 0ull; }
#pragma unroll
        for (int k = 0; k < 4; ++k) wrong += c[k] != e[k];
    }
    if (wrong) atomicAdd(bad, wrong);
}

// one pass's view of the global counters: local bucket owner * sub_n + d = global bucket (owner << log2_sub) + sub_lo + d
__global__ void __launch_bounds__(256)
k_slice(const unsigned long long* __restrict__ acc, uint32_t log2_sub, uint32_t sub_lo, uint32_t sub_n, uint64_t nb_local,
        uint64_t* __restrict__ rec, uint64_t* __restrict__ inst)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nb_local) return;
    const uint64_t owner = j / sub_n, d = j - owner * sub_n;
    unsigned long long h = j < nb_local ? acc[(owner << log2_sub) + sub_lo + d] : 0ull;   // element nb_local = 0: the scans yield totals there
    rec[j] = h >> 32; inst[j] = h & 0xFFFFFFFFull;
}

// items: consecutive fine buckets whose instance prefix falls in the same multiple of `budget`
__global__ void __launch_bounds__(256)
k_item_flags(const uint64_t* __restrict__ ipre, uint64_t nb, uint64_t budget, uint64_t* __restrict__ flags)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    flags[b] = b < nb && (b == 0 || ipre[b] / budget != ipre[b - 1] / budget) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
k_item_pairs(const uint64_t* __restrict__ flags, const uint64_t* __restrict__ idx, uint64_t nb, ItemRange* __restrict__ items)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb || !flags[b]) return;
    const uint64_t k = idx[b];
    items[k].b0 = (uint32_t)b;
    if (k) items[k - 1].b1 = (uint32_t)b;
    if (k + 1 == idx[nb]) items[k].b1 = (uint32_t)nb;
}

// totals of the global counters: out[0] += records, out[1] += instances
__global__ void __launch_bounds__(256)
k_sum_acc(const unsigned long long* __restrict__ acc, uint64_t nb, unsigned long long* __restrict__ out)
{
    unsigned long long r = 0, i = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += (uint64_t)gridDim.x * blockDim.x)
    { r += acc[b] >> 32; i += acc[b] & 0xFFFFFFFFull; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { r += __shfl_down(r, d, 64); i += __shfl_down(i, d, 64); }
    if ((threadIdx.x & 63) == 0) { if (r) atomicAdd(&out[0], r); if (i) atomicAdd(&out[1], i); }
}

__global__ void __launch_bounds__(256)
k_gather_u64(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t n, uint64_t* __restrict__ out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

// every stride-th value of a table of n + 1 (and the last): out[i] = src[min(i * stride, n)], i < n_out
__global__ void __launch_bounds__(256)
k_sample_u64(const uint64_t* __restrict__ src, uint64_t stride, uint64_t n, uint64_t n_out, uint64_t* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_out) out[i] = src[min(i * stride, n)];
}

// ============================================================================ small utilities
// dfk_count with base_off == NULL: reads stored one behind the other, ceil(len/4) bytes each (what BaseVec's feudal writer
// produces) -- the sizes here, their exclusive scan (device_scan, in place) is the offset table
__global__ void __launch_bounds__(256)
k_dense_sizes(const uint32_t* __restrict__ read_len, uint64_t n, uint64_t* __restrict__ out /* [n + 1] */)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * 256) out[i] = i < n ? ((uint64_t)read_len[i] + 3) >> 2 : 0;
}

// DF.cc:447-452 on the device: bc[r] = the barcode whose range [bci[b], bci[b+1]) holds read r (0 for reads no range holds).
// Neighbouring reads take the same path through the index, which stays in the caches.
__global__ void __launch_bounds__(256)
k_expand_bci(const int64_t* __restrict__ bci, uint64_t n_bci, uint64_t n_reads, int32_t* __restrict__ bc)
{
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * 256) {
        // the last b with bci[b] <= r (bci ascending, bci[0] = 0); empty barcodes (bci[b] == bci[b+1]) are stepped over by taking the last
        uint64_t lo = 0, hi = n_bci;                                  // bci[lo] <= r < bci[hi] (hi = n_bci: beyond the table)
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if ((uint64_t)bci[mid] <= r) lo = mid; else hi = mid; }
        bc[r] = (lo + 1 < n_bci && r < (uint64_t)bci[lo + 1]) ? (int32_t)lo : 0;
    }
}

__global__ void k_fill_u64(uint64_t* p, uint64_t n, uint64_t v)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// highest non-zero bin of the spectrum
__global__ void k_hist_max(const unsigned long long* __restrict__ hist, uint32_t n, unsigned int* __restrict__ maxbin)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t stride = gridDim.x * blockDim.x;
    uint32_t best = 0;
    for (; i < n; i += stride) if (hist[i]) best = i + 1;
    if (best) atomicMax(maxbin, best);
}

} // namespace dfk
