/* oracle/dfk_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's k-mer counting hot path
 * (SuperPlus -> vendored Supernova `DF` -> StageBuildGraph -> buildReadQGraph48 ->
 * createDict).  Every function cites the reference
 * file:line it follows (paths relative to /root/reference/lib/assembly/src).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product (libdfk.so) never links or calls it.
 *
 * Pin status: validated against oracle/_ref/refdrv (the reference's own KMer,
 * KMerContext, MapReduceEngine, KmerDict, PQVec and feudal-IO classes compiled in
 * place; glue restated) and against the golden vectors under tests/golden/ that
 * refdrv generated.  The reference ships no tests or golden vectors for this path.
 */
#ifndef DFK_ORACLE_H
#define DFK_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 32-byte KmerDictEntry<K> image (kmers/ReadPather.h:105-146,169-195). */
typedef struct dfko_entry32 {
    uint64_t w0, w1;      /* KMer<K>: bases MSB-first, left-aligned in 128 bits (KMer.h:154-160) */
    uint32_t edge_id;     /* 0xFFFFFFFF = null EdgeID */
    uint32_t count_ctx;   /* count in bits 0-23, KMerContext byte in bits 24-31 */
    int32_t  bc;          /* tempBC; -1 in everything this oracle emits */
    uint32_t pad;
} dfko_entry32;

/* one emitted k-mer instance (Kmerizer::map output, BuildReadQGraph48.cc:148-165) */
typedef struct dfko_inst {
    uint64_t w0, w1;
    int32_t  bc;
    uint32_t ctx;
} dfko_inst;

typedef struct dfko_result {
    uint64_t      n_reads;
    uint32_t*     good_len;      /* [n_reads]  a1 */
    uint64_t      n_inst;        /* a2: number of k-mer instances emitted */
    uint64_t      n_distinct;    /* distinct canonical k-mers */
    uint64_t      n_solid;       /* a4 */
    dfko_entry32* solid_pre;     /* [n_solid] sorted by (w0,w1); contexts BEFORE adjacency (= kmers.kvec content) */
    dfko_entry32* solid;         /* [n_solid] sorted; contexts AFTER recomputeAdjacencies (a6) */
    uint64_t      n_bins;        /* a5: spectrum bins 0..max_count (0 bins if no solid k-mer) */
    int64_t*      hist;
    double        t_trim, t_kmerize, t_count, t_adj;   /* seconds, for the cpu_baseline leg */
} dfko_result;

/* ---- a0 / known answers ---- */
/* KMer<K>(itr) from base codes: KMer.h:154-160. */
void     dfko_kmer_from_codes(const uint8_t* codes, unsigned K, uint64_t w[2]);
/* KMer::hash -> FNV1a over the 16 bytes of (w0,w1): KMer.h:227-230, math/Hash.h:27-35. */
uint64_t dfko_fnv1a16(const uint64_t w[2]);
/* CF<K>::getForm == REV, even K: dna/CanonicalForm.h:58-67. */
int      dfko_is_rev(const uint64_t w[2], unsigned K);
/* KMer::rc: KMer.h:203-225. */
void     dfko_rc(const uint64_t w[2], unsigned K, uint64_t out[2]);
/* KMerContext::rc: KMerContext.h:74-75, KMerContext.cc:19-37 (bit-reversed byte). */
uint8_t  dfko_ctx_rc(uint8_t ctx);

/* ---- formats ---- */
/* PQVecEncoder::decode: feudal/PQVec.cc:129-188.  Returns number of quals, or -1 if the
 * stream runs past nbytes / cap. */
int64_t  dfko_pq_decode(const uint8_t* pq, uint64_t nbytes, uint8_t* q_out, uint64_t cap);
/* A valid (not cost-optimal) PQVec encoding, block layout of PQVecEncoder::encode
 * (feudal/PQVec.cc:87-127).  Returns bytes written (<= n + 3*(n/1+1) ...; give 2*n+8). */
uint64_t dfko_pq_encode(const uint8_t* q, uint32_t n, uint8_t* out);

/* ---- a1 ---- GoodLenTailFinder: BuildReadQGraph48.cc:70-80 */
uint32_t dfko_good_len(const uint8_t* quals, uint32_t n, unsigned K, unsigned min_qual);

/* ---- whole path a1..a6 ----
 * packed_bases/base_off: .fastb var data, 2-bit LSB-first within byte (FieldVec.h:766-770);
 * read_len[n]; pq_bytes/pq_off: .qualp var data; bc[n] or NULL (NULL = no barcode test,
 * BuildReadQGraph{40,60}.cc semantics); ign_bc_below as createDict's ignBcBelow.
 * n_threads <= 0 -> omp default. */
dfko_result* dfko_run(const uint8_t* packed_bases, const uint64_t* base_off, const uint32_t* read_len,
                      const uint8_t* pq_bytes, const uint64_t* pq_off, const int32_t* bc,
                      uint64_t n_reads, unsigned K, unsigned min_qual, unsigned min_freq,
                      unsigned min_bc, int64_t ign_bc_below, int n_threads);
/* same, with good_len supplied (skips a1) */
dfko_result* dfko_run_goodlen(const uint8_t* packed_bases, const uint64_t* base_off,
                      const uint32_t* good_len, const int32_t* bc,
                      uint64_t n_reads, unsigned K, unsigned min_freq,
                      unsigned min_bc, int64_t ign_bc_below, int n_threads);
void dfko_free(dfko_result* r);

/* a2 alone: emits instances in read order into out[cap]; returns count (or needed count if out NULL) */
uint64_t dfko_kmerize(const uint8_t* packed_bases, const uint64_t* base_off, const uint32_t* good_len,
                      const int32_t* bc, int64_t ign_bc_below, uint64_t n_reads, unsigned K,
                      dfko_inst* out, uint64_t cap);

/* a5: exact text of WriteHistToJson<int64_t> (10X/MakeHist.cc:67-92) for "kmer_count"/"DF".
 * Returns bytes needed (excluding NUL); writes at most cap bytes. */
uint64_t dfko_spectrum_json(const int64_t* hist, uint64_t n_bins, char* out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
