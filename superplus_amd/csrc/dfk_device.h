// superplus_amd/csrc/dfk_device.h -- device-side helpers shared by the dfk kernels (gfx950).
//
// Vocabulary:
//   instance      one k-mer occurrence emitted by Kmerizer::map (BuildReadQGraph48.cc:148-165)
//   record        32-byte super-k-mer: a run of <= nk_max consecutive k-mers of one read that
//                 share a minimizer bucket, with one flanking base each side for contexts
//   fine bucket   hash of the run's minimum canonical m-mer hash
//   item          a contiguous range of records (>= 1 fine buckets) counted in one LDS table
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfk {

struct u128 { uint64_t lo, hi; };

__device__ __forceinline__ u128 shr128(u128 x, unsigned s)   // 0 < s < 64
{ return u128{(x.lo >> s) | (x.hi << (64 - s)), x.hi >> s}; }
__device__ __forceinline__ u128 shl128(u128 x, unsigned s)   // 0 < s < 64
{ return u128{x.lo << s, (x.hi << s) | (x.lo >> (64 - s))}; }
__device__ __forceinline__ bool lt128(u128 a, u128 b)
{ return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo; }

// reverse the order of the 64 two-bit groups of x (bit reversal, then swap bits in each pair)
__device__ __forceinline__ uint64_t rev2_64(uint64_t x)
{
    x = __brevll(x);
    return ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);
}

template <int K> struct KTraits {
    static constexpr int BITS = 2 * K;                  // 80 / 96 / 120
    static constexpr int KW = (BITS + 31) / 32;         // key words in the tables: 3 / 3 / 4
    static constexpr int NK_MAX = 95 - K;               // k-mers per record so that nk+K+1 <= 96 bases
    __device__ static __forceinline__ u128 mask()
    { return u128{~0ull, BITS >= 128 ? ~0ull : ((1ull << (BITS - 64)) - 1)}; }
};

// Canonical form of the k-mer whose bases sit, in read order, in the low 2K bits of `ks`
// as little-endian 2-bit fields (base i at bits [2i,2i+2)) -- i.e. exactly as the .fastb
// byte stream stores them (feudal/FieldVec.h:766-770).
//   F = the k-mer as a 2K-bit big-endian number (base 0 most significant; this is KMer<K>'s
//       own order, kmers/KMer.h:154-160) = the 2-bit-group reversal of ks
//   R = its reverse complement in the same order = ~ks (complementing every field of the
//       little-endian stream IS the reverse complement read big-endian)
// CF<K>::getForm for even K (dna/CanonicalForm.h:58-67) walks outside-in and decides at the
// first i with kmer[i] != 3-kmer[K-1-i] = rc[i]; so REV <=> R < F, and a palindrome (R==F)
// is not REV.  Returns min(F,R); *is_rev = R < F.
template <int K>
__device__ __forceinline__ u128 canonical(u128 ks, bool* is_rev)
{
    const u128 m = KTraits<K>::mask();
    u128 R{~ks.lo & m.lo, ~ks.hi & m.hi};
    u128 rv{rev2_64(ks.hi), rev2_64(ks.lo)};            // 128-bit group reversal: fields now top-aligned
    u128 F = shr128(rv, 128 - KTraits<K>::BITS);
    bool rev = lt128(R, F);
    *is_rev = rev;
    return rev ? R : F;
}

// KMerContext::rc (kmers/KMerContext.cc:19-37) = bit reversal of the byte
__device__ __forceinline__ uint32_t ctx_rc(uint32_t c) { return __brev(c) >> 24; }

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
    return x;
}

// Rank of a canonical m-mer among the candidates for minimizer (smaller wins).  The salt keeps poly-A --
// value 0, the most frequent 16-mer of real genomes -- from being the global minimum: mix32(0) = 0.
// The one m-mer that does hash to 0 is the salt itself, AGTACGGTATGCTCAC.
constexpr uint32_t MMER_SALT = 0x2C6B39D1u;
__device__ __forceinline__ uint32_t mmer_hash(uint32_t canon) { return mix32(canon ^ MMER_SALT); }

// slot hash of a canonical k-mer (2K-bit value) for the LDS / HBM tables.  Every key word goes through its own odd
// multiplier BEFORE the words are combined.  (The first version xor-folded rotated words and mixed afterwards: a linear
// fold, under which substitutions in different words cancel.  The k-mers of a diverged repeat family -- one minimizer,
// the other 32 bases a few substitutions away from each other -- then collapsed onto few hash values: 1.67 M distinct
// keys gave 1.25 M hashes, up to 69 keys per value, and at human scale an HBM table of 2^29 slots ran out of its 96
// probe steps at load < 0.5.)
__device__ __forceinline__ uint32_t key_hash(u128 c)
{
    uint32_t a = (uint32_t)c.lo, b = (uint32_t)(c.lo >> 32), d = (uint32_t)c.hi, e = (uint32_t)(c.hi >> 32);
    uint32_t h = (a * 0x9E3779B1u) ^ __builtin_rotateleft32(b * 0x85EBCA77u, 13) ^ __builtin_rotateleft32(d * 0xC2B2AE3Du, 26) ^ (e * 0x27D4EB2Fu);
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15;
    return h;
}

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh)   // ((hi:lo) >> sh) low 32, sh in 0..31
{ return __builtin_amdgcn_alignbit(hi, lo, sh); }

// wave64 inclusive scan on the DPP path: four row_shr steps inside each row of 16 lanes (a source outside the row
// reads 0), then lane 15 of rows 0 and 2 broadcast into rows 1 and 3, then lane 31 into the upper half.  Six VALU
// operations of a few cycles each; as six ds_bpermute shuffles the same scan was ~400 cycles of LDS-crossbar latency in
// the critical path of every staged chunk (twice) and of the finish.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int /*lane*/)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);      // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);      // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
    return (uint32_t)x;
}

// sum over the wave, in every lane (scalar): the scan's last lane
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{ return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0), 63); }

// Lanes of one wave exchanging data through LDS.  The hardware runs a wave's LDS operations in
// order, so no wait is needed, but the COMPILER must be told that other lanes may have written:
// without the fences it forwards a lane's own earlier store to its later load (it did: the
// popcount of a mask word was folded to 0 for lanes that had not set a bit themselves).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A wave-uniform 64-bit value made scalar.  (__builtin_amdgcn_readfirstlane returns int: widened as it stands, a low
// word >= 2^31 sign-extends into the high word -- a record index past 2^31 then addresses memory 64 GiB below its
// block.  Every 64-bit broadcast goes through here.)
__device__ __forceinline__ uint32_t uniform32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v) { return ((uint64_t)uniform32((uint32_t)(v >> 32)) << 32) | (uint64_t)uniform32((uint32_t)v); }
__device__ __forceinline__ uint64_t uniform64(uint32_t lo, uint32_t hi) { return ((uint64_t)uniform32(hi) << 32) | (uint64_t)uniform32(lo); }

// record header (word 0): nk in bits 0-5, has_pred bit 6, has_succ bit 7, fine bucket id in bits 8-31
__device__ __forceinline__ uint32_t rec_header(uint32_t nk, bool hp, bool hs, uint32_t bucket)
{ return nk | (hp ? 64u : 0u) | (hs ? 128u : 0u) | (bucket << 8); }

constexpr uint32_t BCW_MULTI = 0x80000000u;   // barcode word: first barcode seen | MULTI once a second distinct one arrives
// Count word of a table slot: fingerprint (8 bits, never 0) << 24 | count (24 bits).  0 = empty;
// CNT_LOCK (fingerprint 0, count all ones: not a reachable state) = slot being initialised.
constexpr uint32_t CNT_LOCK  = 0x00FFFFFFu;
constexpr uint32_t CNT_MASK  = 0x00FFFFFFu;
constexpr uint32_t CNT_HALF = 0x00800000u;      // from here on the count is bumped by (wave-aggregated) compare-and-swap so that it saturates exactly

} // namespace dfk
