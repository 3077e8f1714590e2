/* oracle/dfk_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See dfk_oracle.h.
 * Plain C restatement of the reference algorithm; clarity first, OpenMP only where it is
 * free.  Paths cited are relative to /root/reference/lib/assembly/src. */
#define _GNU_SOURCE
#include "dfk_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void)
{ struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* ------------------------------------------------------------------ a0: k-mer value type */

/* KMer<K> keeps bases MSB-first in 64-bit words, left-aligned; unused trailing bits are 0
 * (KMer.h:154-160 assign, :327-332 UNUSED_TRAILING_BITS). */
void dfko_kmer_from_codes(const uint8_t* codes, unsigned K, uint64_t w[2])
{
    w[0] = w[1] = 0;
    for (unsigned i = 0; i < K; ++i)
        w[i >> 5] |= (uint64_t)(codes[i] & 3u) << (62 - 2 * (i & 31));
}

static inline unsigned kmer_base(const uint64_t w[2], unsigned i)
{ return (unsigned)(w[i >> 5] >> (62 - 2 * (i & 31))) & 3u; }

/* KMer::hash = FNV1a over the raw bytes of mVal (KMer.h:227-230); x86 is little-endian.
 * FNV-1a 64: math/Hash.h:27-35 (offset basis 14695981039346656037, prime 1099511628211). */
uint64_t dfko_fnv1a16(const uint64_t w[2])
{
    uint64_t h = 14695981039346656037ull;
    for (int k = 0; k < 2; ++k)
        for (int b = 0; b < 8; ++b)
            h = 1099511628211ull * (h ^ ((w[k] >> (8 * b)) & 0xff));
    return h;
}

/* CF<K>::getForm for even K (dna/CanonicalForm.h:58-67): walk outside-in; the first
 * position where base f differs from the complement r of its mirror decides; f<r FWD,
 * r<f REV; none -> PALINDROME (which Kmerizer::map treats as not-REV). */
int dfko_is_rev(const uint64_t w[2], unsigned K)
{
    for (unsigned i = 0, j = K; i < j; ++i) {
        --j;
        unsigned f = kmer_base(w, i), r = kmer_base(w, j) ^ 3u;
        if (f < r) return 0;
        if (r < f) return 1;
    }
    return 0;
}

/* KMer::rc (KMer.h:203-225): base i of the result is the complement of base K-1-i. */
void dfko_rc(const uint64_t w[2], unsigned K, uint64_t out[2])
{
    uint64_t o[2] = {0, 0};
    for (unsigned i = 0; i < K; ++i) {
        unsigned b = kmer_base(w, K - 1 - i) ^ 3u;
        o[i >> 5] |= (uint64_t)b << (62 - 2 * (i & 31));
    }
    out[0] = o[0]; out[1] = o[1];
}

/* KMerContext::rc: the table in KMerContext.cc:19-37 is the 8-bit bit reversal
 * (pred nibble <-> succ nibble, A<->T, C<->G). */
uint8_t dfko_ctx_rc(uint8_t c)
{
    c = (uint8_t)((c >> 4) | (c << 4));
    c = (uint8_t)(((c & 0xcc) >> 2) | ((c & 0x33) << 2));
    c = (uint8_t)(((c & 0xaa) >> 1) | ((c & 0x55) << 1));
    return c;
}

/* KMer::toSuccessor (KMer.h:189-201): shift left one base, new base in the last used slot. */
static inline void kmer_to_successor(uint64_t w[2], unsigned K, unsigned b)
{
    unsigned unused = 128 - 2 * K;
    w[0] = (w[0] << 2) | (w[1] >> 62);
    w[1] = (w[1] << 2) | ((uint64_t)(b & 3u) << unused);
}

/* BaseVec base i: (byte[i/4] >> 2*(i%4)) & 3 (FieldVec.h:766-770,793). */
static inline unsigned read_base(const uint8_t* p, uint64_t i)
{ return (p[i >> 2] >> (2 * (i & 3))) & 3u; }

/* ------------------------------------------------------------------ PQVec codec */

/* Block layout (PQVecEncoder::encode, feudal/PQVec.cc:87-127): byte nQs (0 terminates);
 * then a little-endian bit stream: nBits:3, minQ:6, nQs values of nBits each; padded to a
 * whole byte.  Block size = (nQs*nBits+17+7)>>3 (PQVec.h:58-59).  decode(): PQVec.cc:129-188. */
int64_t dfko_pq_decode(const uint8_t* pq, uint64_t nbytes, uint8_t* q_out, uint64_t cap)
{
    uint64_t pos = 0, n = 0;
    for (;;) {
        if (pos >= nbytes) return -1;
        unsigned nQs = pq[pos];
        if (!nQs) return (int64_t)n;
        if (pos + 3 > nbytes) return -1;
        unsigned hdr = pq[pos + 1] | ((unsigned)pq[pos + 2] << 8);
        unsigned nBits = hdr & 7u, minQ = (hdr >> 3) & 63u;
        uint64_t blk = ((uint64_t)nQs * nBits + 17 + 7) >> 3;
        if (pos + blk > nbytes || n + nQs > cap) return -1;
        uint64_t bit = 8 * (pos + 1) + 9;
        for (unsigned i = 0; i < nQs; ++i) {
            unsigned v = 0;
            for (unsigned b = 0; b < nBits; ++b, ++bit)
                v |= ((pq[bit >> 3] >> (bit & 7)) & 1u) << b;
            q_out[n++] = (uint8_t)(minQ + v);
        }
        pos += blk;
    }
}

static unsigned ceil_lg2(unsigned v) { unsigned b = 0; while ((1u << b) < v) ++b; return b; }

/* Greedy valid encoding: a block is extended while its bit width stays that of its first
 * two values' range (cheap, deterministic).  Same byte layout as encode(); NOT the DP of
 * PQVecEncoder::init (feudal/PQVec.cc:18-85), so bytes differ from the reference encoder
 * while decoding identically. */
uint64_t dfko_pq_encode(const uint8_t* q, uint32_t n, uint8_t* out)
{
    uint64_t o = 0;
    uint32_t i = 0;
    while (i < n) {
        unsigned mn = q[i], mx = q[i];
        uint32_t j = i + 1;
        /* constant runs get nBits=0 blocks; otherwise take up to 255 values */
        while (j < n && j - i < 255 && q[j] == q[i]) ++j;
        if (j - i < 8) {
            j = i + 1;
            while (j < n && j - i < 255) {
                unsigned a = q[j] < mn ? q[j] : mn, b = q[j] > mx ? q[j] : mx;
                /* stop before a long constant run so it can have its own block */
                if (j + 8 <= n) {
                    int c = 1; for (uint32_t t = 1; t < 8; ++t) if (q[j + t] != q[j]) { c = 0; break; }
                    if (c && q[j] != q[j - 1]) break;
                }
                mn = a; mx = b; ++j;
            }
        }
        unsigned nQs = j - i, nBits = ceil_lg2(mx + 1u - mn);
        uint64_t blk = ((uint64_t)nQs * nBits + 17 + 7) >> 3;
        memset(out + o, 0, blk);
        out[o] = (uint8_t)nQs;
        uint64_t bit = 8 * (o + 1);
        unsigned hdr = nBits | (mn << 3);
        for (unsigned b = 0; b < 9; ++b, ++bit) out[bit >> 3] |= (uint8_t)(((hdr >> b) & 1u) << (bit & 7));
        for (uint32_t t = i; t < j; ++t) {
            unsigned v = q[t] - mn;
            for (unsigned b = 0; b < nBits; ++b, ++bit) out[bit >> 3] |= (uint8_t)(((v >> b) & 1u) << (bit & 7));
        }
        o += blk;
        i = j;
    }
    out[o++] = 0;
    return o;
}

/* ------------------------------------------------------------------ a1 */

/* GoodLenTailFinder::operator() (BuildReadQGraph48.cc:70-80): scan from the 3' end; a
 * qual below minQual resets the run; the first time the run reaches K the good length is
 * (index of the run's 5'-most qual) + K; no such run -> 0. */
uint32_t dfko_good_len(const uint8_t* quals, uint32_t n, unsigned K, unsigned min_qual)
{
    unsigned good = 0;
    for (uint32_t i = n; i-- > 0;) {
        if (quals[i] < min_qual) good = 0;
        else if (++good == K) return i + K;
    }
    return 0;
}

/* ------------------------------------------------------------------ a2 */

static inline uint64_t inst_of_len(uint32_t g, unsigned K) { return g >= K + 1 ? (uint64_t)g - K + 1 : 0; }

/* Kmerizer::map (BuildReadQGraph48.cc:148-165) for one read. */
static uint64_t kmerize_read(const uint8_t* p, uint32_t len, int32_t tag, unsigned K, dfko_inst* out)
{
    if (len < K + 1) return 0;                             /* :153 */
    uint64_t w[2] = {0, 0};
    for (unsigned i = 0; i < K; ++i)
        w[i >> 5] |= (uint64_t)read_base(p, i) << (62 - 2 * (i & 31));
    uint64_t n = 0;
    for (uint32_t j = 0; j + K <= len; ++j) {
        if (j) kmer_to_successor(w, K, read_base(p, j + K - 1));
        uint8_t ctx = 0;
        if (j > 0)       ctx |= (uint8_t)(0x10u << read_base(p, j - 1));   /* predecessor (KMerContext.h:91-95) */
        if (j + K < len) ctx |= (uint8_t)(0x01u << read_base(p, j + K));   /* successor; last k-mer has none (:163) */
        dfko_inst* e = &out[n++];
        if (dfko_is_rev(w, K)) { dfko_rc(w, K, &e->w0); e->ctx = dfko_ctx_rc(ctx); }   /* :156,161,165 */
        else { e->w0 = w[0]; e->w1 = w[1]; e->ctx = ctx; }
        e->bc = tag;
    }
    return n;
}

uint64_t dfko_kmerize(const uint8_t* packed, const uint64_t* base_off, const uint32_t* good_len,
                      const int32_t* bc, int64_t ign_bc_below, uint64_t n_reads, unsigned K,
                      dfko_inst* out, uint64_t cap)
{
    uint64_t total = 0;
    for (uint64_t r = 0; r < n_reads; ++r) total += inst_of_len(good_len[r], K);
    if (!out) return total;
    if (cap < total) return total;
    uint64_t* start = (uint64_t*)malloc(sizeof(uint64_t) * (n_reads + 1));
    start[0] = 0;
    for (uint64_t r = 0; r < n_reads; ++r) start[r + 1] = start[r] + inst_of_len(good_len[r], K);
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
        int32_t tag = -1;                                  /* :150-151 */
        if (bc && r >= ign_bc_below) tag = bc[r];
        kmerize_read(packed + base_off[r], good_len[r], tag, K, out + start[r]);
    }
    free(start);
    return total;
}

/* ------------------------------------------------------------------ a3 + a4 + a5 */

static int inst_cmp(const void* a, const void* b)
{
    const dfko_inst* x = (const dfko_inst*)a; const dfko_inst* y = (const dfko_inst*)b;
    if (x->w0 != y->w0) return x->w0 < y->w0 ? -1 : 1;    /* KMer operator< : KMer.h:287-311 */
    if (x->w1 != y->w1) return x->w1 < y->w1 ? -1 : 1;
    return 0;
}

/* Kmerizer::reduce (BuildReadQGraph48.cc:167-174) on one run of equal k-mers:
 * summarizeEntries (:89-102) ORs contexts and adds counts (saturating at 2^24-1 through
 * KDef::setCount, ReadPather.h:128-129); barcode test = areIgnoredBarcodes (:104-110) ||
 * areEnoughBarcodes (:112-132: distinct barcodes > 0 reach minBC). */
static int reduce_run(const dfko_inst* a, const dfko_inst* b, unsigned min_freq, unsigned min_bc,
                      int use_bc, dfko_entry32* out)
{
    uint32_t ctx = 0; uint64_t cnt = 0;
    for (const dfko_inst* e = a; e != b; ++e) { ctx |= e->ctx; ++cnt; }
    if (cnt > 0xFFFFFFull) cnt = 0xFFFFFFull;
    int ok = 1;
    if (use_bc) {
        ok = 0;
        for (const dfko_inst* e = a; e != b; ++e) if (e->bc == -1) { ok = 1; break; }
        if (!ok) {
            unsigned distinct = 0;
            int32_t small[16];
            int32_t* seen = min_bc <= 16 ? small : (int32_t*)malloc(sizeof(int32_t) * min_bc);
            if (min_bc == 0) ok = 1;
            for (const dfko_inst* e = a; e != b && !ok; ++e) {
                if (e->bc <= 0) continue;                   /* unset (-1) and barcode 0 never count (:122) */
                int dup = 0;
                for (unsigned t = 0; t < distinct; ++t) if (seen[t] == e->bc) { dup = 1; break; }
                if (dup) continue;
                seen[distinct++] = e->bc;
                if (distinct >= min_bc) ok = 1;
            }
            if (seen != small) free(seen);
        }
    }
    if (!(cnt >= min_freq && ok)) return 0;
    out->w0 = a->w0; out->w1 = a->w1; out->edge_id = 0xFFFFFFFFu;
    out->count_ctx = (uint32_t)cnt | (ctx << 24);
    out->bc = -1; out->pad = 0;
    return 1;
}

#define NPART 4096   /* order-preserving partition on the top 12 bits of w0 */

/* ------------------------------------------------------------------ a6 */

static int64_t find_key(const dfko_entry32* s, uint64_t n, uint64_t w0, uint64_t w1)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (s[mid].w0 < w0 || (s[mid].w0 == w0 && s[mid].w1 < w1)) lo = mid + 1; else hi = mid;
    }
    return (lo < n && s[lo].w0 == w0 && s[lo].w1 == w1) ? (int64_t)lo : -1;
}

static int solid_has(const dfko_entry32* s, uint64_t n, const uint64_t w[2], unsigned K)
{   /* KmerDict::findEntry canonicalises first (ReadPather.h:222-225) */
    uint64_t c[2] = {w[0], w[1]};
    if (dfko_is_rev(w, K)) dfko_rc(w, K, c);
    return find_key(s, n, c[0], c[1]) >= 0;
}

/* KmerDict::recomputeAdjacencies (ReadPather.h:329-364): for every set successor bit b,
 * keep it only if canonical(kmer[1:]+b) is in the dictionary; likewise predecessors with
 * b+kmer[:-1].  Tests presence in the (fixed) solid set only, so order cannot matter. */
static void adjacency(const dfko_entry32* pre, dfko_entry32* post, uint64_t n, unsigned K)
{
    unsigned unused = 128 - 2 * K;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        uint32_t ctx = pre[i].count_ctx >> 24;
        uint64_t k[2] = {pre[i].w0, pre[i].w1};
        if (ctx & 0x0f) {
            uint64_t s[2] = {k[0], k[1]};
            kmer_to_successor(s, K, 0);
            for (unsigned b = 0; b < 4; ++b) if (ctx & (1u << b)) {
                uint64_t t[2] = {s[0], s[1] | ((uint64_t)b << unused)};
                if (!solid_has(pre, n, t, K)) ctx &= ~(1u << b);
            }
        }
        if (ctx & 0xf0) {
            /* KMer::toPredecessor (KMer.h:175-187): shift right one base, new base in front */
            uint64_t p[2];
            p[1] = ((k[1] >> 2) | (k[0] << 62)) & ~((unused ? ((uint64_t)1 << unused) : 1) - 1);
            p[0] = k[0] >> 2;
            for (unsigned b = 0; b < 4; ++b) if (ctx & (0x10u << b)) {
                uint64_t t[2] = {p[0] | ((uint64_t)b << 62), p[1]};
                if (!solid_has(pre, n, t, K)) ctx &= ~(0x10u << b);
            }
        }
        post[i] = pre[i];
        post[i].count_ctx = (pre[i].count_ctx & 0xFFFFFFu) | (ctx << 24);
    }
}

/* ------------------------------------------------------------------ driver */

dfko_result* dfko_run_goodlen(const uint8_t* packed, const uint64_t* base_off,
                      const uint32_t* good_len, const int32_t* bc,
                      uint64_t n_reads, unsigned K, unsigned min_freq,
                      unsigned min_bc, int64_t ign_bc_below, int n_threads)
{
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    dfko_result* R = (dfko_result*)calloc(1, sizeof(dfko_result));
    R->n_reads = n_reads;
    R->good_len = (uint32_t*)malloc(sizeof(uint32_t) * (n_reads ? n_reads : 1));
    memcpy(R->good_len, good_len, sizeof(uint32_t) * n_reads);

    double t0 = now_s();
    uint64_t n_inst = dfko_kmerize(packed, base_off, good_len, bc, ign_bc_below, n_reads, K, NULL, 0);
    dfko_inst* inst = (dfko_inst*)malloc(sizeof(dfko_inst) * (n_inst ? n_inst : 1));
    dfko_kmerize(packed, base_off, good_len, bc, ign_bc_below, n_reads, K, inst, n_inst);
    R->n_inst = n_inst;
    double t1 = now_s();

    /* MapReduceEngine (MapReduceEngine.h:313-388,570-580) hash-partitions, sorts each
     * partition by k-mer and reduces equal runs.  The grouping, not the partition
     * function, defines the result; partition here on the top bits of w0 so that the
     * concatenated output is already sorted. */
    uint64_t* pstart = (uint64_t*)calloc(NPART + 1, sizeof(uint64_t));
    for (uint64_t i = 0; i < n_inst; ++i) pstart[(inst[i].w0 >> 52) + 1]++;
    for (int p = 0; p < NPART; ++p) pstart[p + 1] += pstart[p];
    dfko_inst* part = (dfko_inst*)malloc(sizeof(dfko_inst) * (n_inst ? n_inst : 1));
    { uint64_t* cur = (uint64_t*)malloc(sizeof(uint64_t) * NPART);
      memcpy(cur, pstart, sizeof(uint64_t) * NPART);
      for (uint64_t i = 0; i < n_inst; ++i) part[cur[inst[i].w0 >> 52]++] = inst[i];
      free(cur); }
    free(inst);
    uint64_t* nsol = (uint64_t*)calloc(NPART + 1, sizeof(uint64_t));
    uint64_t* ndis = (uint64_t*)calloc(NPART, sizeof(uint64_t));
    dfko_entry32** pout = (dfko_entry32**)calloc(NPART, sizeof(dfko_entry32*));
    int use_bc = bc != NULL;
#pragma omp parallel for schedule(dynamic, 1)
    for (int p = 0; p < NPART; ++p) {
        uint64_t a = pstart[p], b = pstart[p + 1];
        if (a == b) continue;
        qsort(part + a, b - a, sizeof(dfko_inst), inst_cmp);
        dfko_entry32* o = (dfko_entry32*)malloc(sizeof(dfko_entry32) * (b - a));
        uint64_t no = 0, nd = 0;
        for (uint64_t i = a; i < b;) {
            uint64_t j = i + 1;
            while (j < b && part[j].w0 == part[i].w0 && part[j].w1 == part[i].w1) ++j;
            no += reduce_run(part + i, part + j, min_freq, min_bc, use_bc, o + no);
            ++nd; i = j;
        }
        pout[p] = o; nsol[p + 1] = no; ndis[p] = nd;
    }
    for (int p = 0; p < NPART; ++p) { nsol[p + 1] += nsol[p]; R->n_distinct += ndis[p]; }
    R->n_solid = nsol[NPART];
    R->solid_pre = (dfko_entry32*)malloc(sizeof(dfko_entry32) * (R->n_solid ? R->n_solid : 1));
    for (int p = 0; p < NPART; ++p) if (pout[p]) {
        memcpy(R->solid_pre + nsol[p], pout[p], sizeof(dfko_entry32) * (nsol[p + 1] - nsol[p]));
        free(pout[p]);
    }
    free(pout); free(nsol); free(ndis); free(part); free(pstart);

    /* WriteKmerSpectrum (BuildReadQGraph48.cc:192-209): hist[count]++, trailing zeros pruned. */
    uint64_t maxc = 0; int any = 0;
    for (uint64_t i = 0; i < R->n_solid; ++i) { uint64_t c = R->solid_pre[i].count_ctx & 0xFFFFFFu; if (c > maxc) maxc = c; any = 1; }
    R->n_bins = any ? maxc + 1 : 0;
    R->hist = (int64_t*)calloc(R->n_bins ? R->n_bins : 1, sizeof(int64_t));
    for (uint64_t i = 0; i < R->n_solid; ++i) R->hist[R->solid_pre[i].count_ctx & 0xFFFFFFu]++;
    double t2 = now_s();

    R->solid = (dfko_entry32*)malloc(sizeof(dfko_entry32) * (R->n_solid ? R->n_solid : 1));
    if (min_freq > 1) adjacency(R->solid_pre, R->solid, R->n_solid, K);      /* :312-314 */
    else memcpy(R->solid, R->solid_pre, sizeof(dfko_entry32) * R->n_solid);
    double t3 = now_s();
    R->t_kmerize = t1 - t0; R->t_count = t2 - t1; R->t_adj = t3 - t2;
    return R;
}

dfko_result* dfko_run(const uint8_t* packed, const uint64_t* base_off, const uint32_t* read_len,
                      const uint8_t* pq, const uint64_t* pq_off, const int32_t* bc,
                      uint64_t n_reads, unsigned K, unsigned min_qual, unsigned min_freq,
                      unsigned min_bc, int64_t ign_bc_below, int n_threads)
{
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    double t0 = now_s();
    uint32_t* gl = (uint32_t*)malloc(sizeof(uint32_t) * (n_reads ? n_reads : 1));
    int bad = 0;
#pragma omp parallel
    {
        uint8_t* q = (uint8_t*)malloc(65536);
#pragma omp for schedule(dynamic, 4096)
        for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
            int64_t n = dfko_pq_decode(pq + pq_off[r], pq_off[r + 1] - pq_off[r], q, 65536);
            if (n < 0 || (uint64_t)n != read_len[r]) { bad = 1; gl[r] = 0; continue; }
            gl[r] = dfko_good_len(q, (uint32_t)n, K, min_qual);
        }
        free(q);
    }
    double t1 = now_s();
    if (bad) { free(gl); return NULL; }
    dfko_result* R = dfko_run_goodlen(packed, base_off, gl, bc, n_reads, K, min_freq, min_bc, ign_bc_below, n_threads);
    R->t_trim = t1 - t0;
    free(gl);
    return R;
}

void dfko_free(dfko_result* r)
{
    if (!r) return;
    free(r->good_len); free(r->solid_pre); free(r->solid); free(r->hist); free(r);
}

/* WriteHistToJson<int64_t> (10X/MakeHist.cc:67-92), called as
 * WriteHistToJson(kmerspec, 0, max_count, 1, dir, "kmer_count", "DF") (BuildReadQGraph48.cc:208). */
uint64_t dfko_spectrum_json(const int64_t* hist, uint64_t n_bins, char* out, uint64_t cap)
{
    uint64_t need = 0;
    char tmp[256];
#define EMIT(...) do { int k_ = snprintf(tmp, sizeof tmp, __VA_ARGS__); \
        if (out && need + (uint64_t)k_ <= cap) memcpy(out + need, tmp, (size_t)k_); need += (uint64_t)k_; } while (0)
    EMIT("{\n\t\"description\": \"kmer_count\",\n\t\"stage\": \"DF\",\n\t\"binsize\": 1,\n\t\"min\": 0,\n");
    EMIT("\t\"max\": %lld,\n\t\"numbins\": %llu,\n\t\"vals\": [", (long long)n_bins - 1, (unsigned long long)n_bins);
    for (uint64_t i = 0; i < n_bins; ++i) { EMIT("%lld", (long long)hist[i]); if (i + 1 != n_bins) EMIT(","); }
    EMIT("]\n}\n");
#undef EMIT
    if (out && need < cap) out[need] = 0;
    return need;
}
