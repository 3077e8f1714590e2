/* include/dfk.h -- C ABI of libdfk.so: the MI355X-native replacement for the k-mer counting
 * hot path that SuperPlus runs as the vendored Supernova `DF` binary.
 *
 * The reference has no plugin/FFI seam for this path; the seam it does have is one
 * file-static C++ call (all reference paths relative to lib/assembly/src):
 *
 *   Dict* createDict(String const& work_dir, vecbvec const& reads,
 *                    ObjectManager<VecPQVec>& quals, unsigned minQual, unsigned minFreq,
 *                    int64_t ignBcBelow, float mem_frac, unsigned minBC,
 *                    vec<int32_t> const* bcp)          paths/long/BuildReadQGraph48.cc:211-215
 *   called from buildReadQGraph48()                    paths/long/BuildReadQGraph48.cc:1626
 *   called from StageBuildGraph()                      10X/runstages/RunStages.cc:389
 *
 * Each entry point below names the part of that call it replaces.  Plain pointers and
 * sizes only; no exceptions cross the boundary; every function returns 0 on success or a
 * negative DFK_E_* code, with a message available from dfk_last_error().  One context per
 * GPU; calls on one context must be serialised by the caller (createDict is likewise
 * called once, from the main thread: MapReduceEngine.h:423-427).
 *
 * There is NO CPU fallback: if no gfx950 device is usable, dfk_create() fails.
 */
#ifndef DFK_H
#define DFK_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DFK_ABI_VERSION 1
#define DFK_MAX_MIN_BC 8u

enum {
    DFK_OK            = 0,
    DFK_E_ARG         = -1,   /* bad argument / unsupported configuration */
    DFK_E_NODEVICE    = -2,   /* no usable HIP device (the product has no CPU path) */
    DFK_E_HIP         = -3,   /* a HIP runtime call failed */
    DFK_E_NOMEM       = -4,   /* does not fit the HBM budget */
    DFK_E_INPUT       = -5,   /* malformed input (e.g. PQVec length != read length) */
    DFK_E_STATE       = -6,   /* call out of order (e.g. fetch before count) */
    DFK_E_NOGOOD      = -7    /* "almost no good bases": createDict's Scram(1), BuildReadQGraph48.cc:227-230 */
};

/* Mirrors createDict's scalar arguments.  K: the reference instantiates 40, 48 and 60
 * (BuildReadQGraph{40,48,60}.cc); DF's CLI only reaches 48 (RunStages.cc:388). */
typedef struct dfk_config {
    uint32_t abi_version;       /* DFK_ABI_VERSION */
    uint32_t K;                 /* 40, 48 or 60 */
    uint32_t min_qual;          /* MIN_QUAL, default 7   (10X/DF.cc:129-132) */
    uint32_t min_freq;          /* MIN_FREQ, default 3 */
    uint32_t min_bc;            /* MIN_BC,   default 2; 0..DFK_MAX_MIN_BC (a table slot remembers MIN_BC-1 distinct barcodes) */
    int32_t  device;            /* HIP device ordinal */
    int64_t  ign_bc_below;      /* createDict ignBcBelow (= bc_start, DF.cc:344-349) */
    uint64_t hbm_budget_bytes;  /* 0 = 90 % of free HBM (mem_frac analogue, GRAPHMEM=0.9); larger requests are clamped to that */
    uint32_t minimizer_len;     /* 0 = default; 8..16 */
    uint32_t flags;             /* DFK_F_* */
    uint64_t inst_per_item;     /* 0 = default; k-mer instances packed into one LDS table pass */
    uint64_t reserved[4];       /* [0] = forced number of bucket-range passes, 0 = sized from the free HBM */
} dfk_config;

#define DFK_F_KEEP_PRE_ADJ   1u   /* also keep contexts before recomputeAdjacencies (kmers.kvec view) */
#define DFK_F_KEEP_INPUTS    2u   /* dfk_count keeps its device copies of the reads (for dfk_paths_build) until the next count */

/* 32-byte image of KmerDictEntry<K> (kmers/ReadPather.h:105-146,169-195):
 * w0 = bases 0..31 MSB-first, w1 = remaining bases left-aligned (kmers/KMer.h:154-160),
 * edge_id = 0xFFFFFFFF (null), count in bits 0-23 and KMerContext byte in bits 24-31 of
 * count_ctx, tempBC = -1, pad = 0. */
typedef struct dfk_entry32 {
    uint64_t w0, w1;
    uint32_t edge_id;
    uint32_t count_ctx;
    int32_t  bc;
    uint32_t pad;
} dfk_entry32;

typedef struct dfk_stats {
    uint64_t n_reads;
    uint64_t n_inst;            /* k-mer instances Kmerizer::map would emit (the metric's unit) */
    uint64_t n_records;         /* super-k-mer records written by the partition kernel */
    uint64_t n_buckets;         /* fine minimizer buckets */
    uint64_t n_items;           /* LDS table passes */
    uint64_t n_overflow_items;  /* items that had to be split or re-run in the HBM table */
    uint64_t n_distinct;        /* distinct canonical k-mers seen */
    uint64_t n_solid;
    uint64_t adj_probes;        /* neighbour look-ups issued by the adjacency kernel */
    /* GPU time of each stage of the last dfk_count*, milliseconds, from HIP events on the
     * context's stream */
    float ms_upload, ms_trim, ms_part_count, ms_part_scatter, ms_count, ms_fallback, ms_adjacency, ms_total;
    uint64_t hbm_bytes_peak;    /* peak device bytes held by the context */
    uint64_t reserved[8];       /* [0] = passes used, [1..3] = microseconds (graph device, graph host, pathing), [4] = gate timeouts, [5] = device bytes held now, [6] = launches of the counting scan */
} dfk_stats;

typedef struct dfk_ctx dfk_ctx;

/* Construct / destroy.  Replaces nothing in the reference (its state lives on createDict's stack). */
int         dfk_create(const dfk_config* cfg, dfk_ctx** out);
void        dfk_destroy(dfk_ctx* ctx);
const char* dfk_last_error(void);
int         dfk_abi_version(void);

/* The quality histogram of DF's side file frag_reads_orig.qhist (10X/DF.cc:50-68, DfTools.cc:172-238) from the reads
 * dfk_count kept on the device (DFK_F_KEEP_INPUTS): hist[parity][pos][q], parity = read index & 1, pos < max_len, q < 256 --
 * 2 * max_len * 256 counts.  Positions from max_len on are left out. */
int dfk_qual_hist(dfk_ctx* ctx, uint32_t max_len, int64_t* hist);

/* A hint for the host-buffer calls (dfk_count, dfk_shard_begin_host, dfk_paths_build): the host range
 * [base, base + bytes) is a shared, read-only mapping of a file.  Unmapping a 90-GB input every page of which was touched
 * is three seconds of page-table work for one thread -- at exit, if not before; with the hint it is not left to one thread:
 *   fd >= 0  the mapping shows the file from `file_off` on, and the library READS THE FILE (pread) for every byte of an
 *            input array inside the range instead of the memory: pages nobody else touched are never mapped, and the caller
 *            may unmap the range while the call runs (the descriptor must stay open).  A failing read is DFK_E_INPUT,
 *            never a quiet fall back to the memory.  (Measured: 15 GB/s whatever the number of lanes.)
 *   fd = -1  the library reads the memory and DROPS what it has read from the caller's page table (MADV_DONTNEED: the file
 *            keeps the pages; a later access faults them in again) -- each transfer lane its own chunks -- and marks the
 *            mapping MADV_SEQUENTIAL, without which every page that leaves a page table is marked accessed under one lock
 *            (a freshly written tmpfs file: 25 against 130 GB/s for its first reading).  The mapping must stay until the
 *            call returns.  (Measured: 40-50 GB/s.)
 * base = NULL forgets all hints; at most 8 are kept; dfk_destroy forgets them.  Replaces nothing in the reference (LoadData
 * reads its files through read(2) into vectors, feudal/FeudalFileReader.cc). */
int dfk_hint_file_range(dfk_ctx* ctx, const void* base, uint64_t bytes, int fd, uint64_t file_off);

/* createDict(...) on host buffers laid out exactly as the .fastb/.qualp var data and DF's
 * expanded barcode vector (DF.cc:447-452):
 *   packed_bases + base_off[n+1]  BaseVec bytes, 2-bit LSB-first (feudal/FieldVec.h:766-770)
 *   read_len[n]                   bases per read
 *   pq_bytes + pq_off[n+1]        PQVec block streams (feudal/PQVec.cc:87-127)
 *   bc[n] or NULL                 per-read barcode id; NULL = no barcode test at all
 *                                 (the K=40/60 variants, BuildReadQGraph60.cc:103-151)
 * Runs: quality-tail trim (:218-225), canonical k-mer extraction with contexts (:148-165),
 * counting and the MIN_FREQ/MIN_BC solid filter (:167-174), the spectrum (:192-209) and
 * recomputeAdjacencies (ReadPather.h:329-364, gated by minFreq > 1 as at :313). */
/* Input contract (both calls): the offset tables are checked on the device before anything is read through them
 * (monotone, last offset <= the array size, base_off[r+1]-base_off[r] >= ceil(read_len[r]/4)) and a violation is
 * DFK_E_INPUT.  packed_bases need not start at the first read (a mapped .fastb file with its absolute offset table
 * is a valid input).  Device arrays: packed_bases 4-byte aligned and readable up to 3 bytes past packed_bytes (the
 * 2-bit stream is read as aligned 32-bit words); the offset tables 8-byte aligned.
 * From host buffers the qualities and tables are uploaded first and the bases last, in pieces (from 1 GB of bases on): the
 * trim runs beside the first piece and the counting scan follows the pieces, so that upload and count overlap by what the
 * two take (0.4 s at configs[1]); the result does not depend on it (dfk_stats.reserved[6]: launches of the scan). */
int dfk_count(dfk_ctx* ctx,
              const uint8_t* packed_bases, const uint64_t* base_off, const uint32_t* read_len,
              const uint8_t* pq_bytes, const uint64_t* pq_off, const int32_t* bc,
              uint64_t n_reads);
/* What need not cross PCIe (DF, 10X/DF.cc:300-452, holds both in this form when it reaches createDict):
 *   base_off == NULL (either call)  the bases are DENSE: read r starts where read r-1 ended, ceil(read_len/4) bytes each, read 0
 *                                   at packed_bases -- what BaseVec's feudal writer produces (feudal/FeudalFileWriter.cc:100-121).
 *                                   The offset table is derived on the device from read_len (8 bytes a read not uploaded).
 *   dfk_count_bci                   takes the barcode INDEX of head.bci / frag_reads_orig.bci (n_bci entries, ascending, bci[0] = 0:
 *                                   reads [bci[b], bci[b+1]) carry barcode b) in place of the vector DF expands it to at
 *                                   10X/DF.cc:447-452; the expansion runs on the device (4 bytes a read not uploaded, and not
 *                                   written by the host first).  Reads beyond bci[n_bci-1] are unbarcoded (0). */
int dfk_count_bci(dfk_ctx* ctx,
              const uint8_t* packed_bases, const uint64_t* base_off, const uint32_t* read_len,
              const uint8_t* pq_bytes, const uint64_t* pq_off, const int64_t* bci, uint64_t n_bci,
              uint64_t n_reads);

/* Same call with every array already resident in this context's device memory (bench.py's
 * timed region; multi-GPU shards).  packed_bytes / pq_nbytes are the allocation sizes. */
int dfk_count_device(dfk_ctx* ctx,
              const void* d_packed_bases, uint64_t packed_bytes, const void* d_base_off,
              const void* d_read_len, const void* d_pq_bytes, uint64_t pq_nbytes,
              const void* d_pq_off, const void* d_bc, uint64_t n_reads);

/* goodLens (BuildReadQGraph48.cc:218-225) of the last count, for parity tests. */
int dfk_good_lens(dfk_ctx* ctx, uint32_t* out, uint64_t cap);

/* WriteKmerSpectrum's vector (BuildReadQGraph48.cc:192-209): hist[c] = #solid k-mers with
 * count c, c in [0,max]; *nbins = max+1 (0 when there is no solid k-mer).  The pointer
 * stays valid until the next dfk_count* or dfk_destroy. */
int dfk_spectrum(dfk_ctx* ctx, const int64_t** hist, uint64_t* nbins);
/* Exact text of stats/histogram_kmer_count.json (10X/MakeHist.cc:67-92).  Returns bytes
 * needed (excl. NUL) through *need; writes at most cap bytes. */
int dfk_spectrum_json(dfk_ctx* ctx, char* out, uint64_t cap, uint64_t* need);

/* The Dict's content: number of solid k-mers, then the entries sorted ascending by
 * (w0,w1) with contexts AFTER recomputeAdjacencies.  pre_adjacency != 0 returns the
 * kmers.kvec view (contexts before; needs DFK_F_KEEP_PRE_ADJ). */
int dfk_solid_count(dfk_ctx* ctx, uint64_t* n);
int dfk_solid_fetch(dfk_ctx* ctx, dfk_entry32* out, uint64_t cap, int pre_adjacency);
/* The same entries in the order the device holds them (one copy per pass, no sort): what a caller that
 * inserts them into its own dictionary needs -- the reference's kmers.kvec is in thread-arrival order too. */
int dfk_solid_fetch_unsorted(dfk_ctx* ctx, dfk_entry32* out, uint64_t cap, int pre_adjacency);

/* Order-independent digest of the same entries, computed on the device: digest[0] = sum over entries of h(entry)
 * mod 2^64, digest[1] = xor of h'(entry), h = a 64-bit mix of the entry's 32 bytes (k_digest in dfk_kernels.h;
 * tests/util.py has the numpy form).  Replaces nothing in the reference: it is how two dictionaries of billions
 * of entries are compared without fetching them (pass geometries, single GPU against sharded: digests of the
 * ranks' disjoint shares add / xor). */
int dfk_solid_digest(dfk_ctx* ctx, int pre_adjacency, uint64_t* digest /* [2] */);

/* kmers.kvec image ("BINWRITE" | u64 n | n x 32-B entries; BuildReadQGraph48.cc:287-288,
 * feudal/BinaryStream.h:33-46) written straight to a file, streamed from the device in the order the
 * device holds the entries -- the reference's own file is in thread-arrival order (ReadPather.h:406-418)
 * and its reader inserts the entries into a hash set (BuildReadQGraph48.cc:294-301), so no order is
 * promised by either side.  flags: DFK_KVEC_PRE_ADJ = contexts before recomputeAdjacencies (needs
 * DFK_F_KEEP_PRE_ADJ); DFK_KVEC_SORTED = ascending (w0,w1), sorted on the host (parity tests: needs two
 * host copies of the dictionary). */
#define DFK_KVEC_PRE_ADJ 1
#define DFK_KVEC_SORTED  2
int dfk_write_kvec(dfk_ctx* ctx, const char* path, int flags);
/* The same file written by several ranks (multi-GPU runs: solid sets are disjoint): every rank writes its share at
 * entry first_entry of a file of total_entries entries (the caller derives first_entry from the ranks' solid counts);
 * the rank with first_entry == 0 writes the header.  Device order only. */
int dfk_write_kvec_part(dfk_ctx* ctx, const char* path, int flags, uint64_t first_entry, uint64_t total_entries);

int dfk_get_stats(dfk_ctx* ctx, dfk_stats* out);

/* ---- the graph of the solid k-mers (single-GPU counts) ----
 * dfk_graph_build replaces, on the dictionary of the last dfk_count*:
 *   buildEdges         (paths/long/BuildReadQGraph48.cc:505-530, EdgeBuilder :320-503): the unipath edges, each once
 *                      in canonical form -- on the device; as in the reference every dictionary entry then carries
 *                      (edge id, offset on the edge) in place of (null, count) (KDef::set, kmers/ReadPather.h:122-127),
 *                      so dfk_solid_fetch afterwards returns offsets where it returned counts;
 *   buildHBVFromEdges  (paths/long/HBVFromEdges.cc:244-296): vertices = the (K-1)-mers at the edge ends, edges numbered
 *                      in the reference's canonical traversal order, both orientations of every edge -- on the host
 *                      (the numbering IS a sequential traversal).
 * dfk_graph_write writes what WriteAssemblyFiles (10X/WriteFiles.cc:69-101) writes of a.<K>/ for the graph itself:
 * a.k, a.hbv (K | digraphE<basevector>), a.hbx (HyperBasevectorX), a.to_left, a.to_right, a.inv
 * (HyperBasevector::Involution), a.fastb (the edges), a.kmers -- byte for byte; the paths files need the read pather
 * (SURVEY 8(f)-2).  dfk_stats.reserved[1] / [2]: microseconds spent on the edges (device) / the numbering (host). */
int dfk_graph_build(dfk_ctx* ctx);
int dfk_graph_stats(dfk_ctx* ctx, uint64_t* n_canonical_edges, uint64_t* n_vertices, uint64_t* n_edges);
int dfk_graph_write(dfk_ctx* ctx, const char* dir);

/* a.paths written WHILE the reads are being pathed: the next dfk_paths_build creates `path` and a thread of the library's
 * takes every finished batch's variable data to its place in the file; dfk_paths_write(ctx, path) -- the same path -- then
 * waits for that thread and adds what only the whole build knows (control block, offset table).  One file on a tmpfs takes
 * what one writer gives (9 GB/s): this starts the 37 GB of configs[1] two seconds earlier.  path = NULL: back to
 * dfk_paths_write writing everything.  A build that fails removes the file.  (ReadPathVec::WriteAll, 10X/WriteFiles.cc:78-82,
 * writes after pathReads has returned.) */
/* A file that exists at `path` is not truncated when the build starts (dfk_paths_write sets its size): a caller that has made it
 * about as large as a.paths will be (fallocate of 20-odd bytes a read, while the GPU counts) saves the writer the allocation
 * of every page under the file's lock.  The same holds for dir/a.paths.inv of dfk_paths_index_write. */
int dfk_paths_sink(dfk_ctx* ctx, const char* path);

/* ---- read pathing ("correction by pathing") on the graph dfk_graph_build left in the context ----
 * dfk_paths_build replaces pathReads (paths/long/BuildReadQGraph48.cc:1420-1442) as buildReadQGraph48 calls it
 * (:1664-1665, with useNewAligner = True from StageBuildGraph, 10X/runstages/RunStages.cc:389-390): every read --
 * whole, not quality-trimmed -- is looked up k-mer by k-mer in the dictionary and followed along the unipath edges
 * (Pather::path, :685-733), the parts are edited by the rules of HBVPather::algorithmTwo (:1212-1317) and the path is
 * extended left and right over read ends that hang beyond it by the quality-weighted scores of ExtendReadPath
 * (paths/long/ExtendReadPath.cc).  One ReadPath per read: an offset on its first edge and HBV edge ids.
 *   dfk_paths_build         reads in host memory, laid out as for dfk_count (uploaded, pathed, freed); all five arrays
 *                           NULL = the reads the last dfk_count uploaded and kept (DFK_F_KEEP_INPUTS)
 *   dfk_paths_build_device  the same arrays already resident on the context's device
 *   dfk_paths_write         a.<K>/a.paths as WriteAssemblyFiles writes it (10X/WriteFiles.cc:78-82: ReadPathVec::WriteAll, a feudal
 *                           file of {i32 offset, u32 lastSkip = 0, i32 edges...}, paths/long/ReadPath.h:56-63) -- byte for byte
 *   dfk_paths_fetch         offsets[n], first_edge[n+1] (read r's edges are edges[first_edge[r] .. first_edge[r+1])), edges
 * Needs dfk_graph_build on the same count.  dfk_stats.reserved[3]: microseconds spent pathing.
 * dfk_graph_write and dfk_paths_write only READ the context (host tables, finished device buffers) and allocate nothing on the
 * device: each may run on a thread of its own beside the NEXT call in the chain -- dfk_graph_write beside dfk_paths_build,
 * dfk_paths_write beside dfk_paths_index_write / dfk_dups_write -- which is how DF hides the files under the device work.
 * Every other call on a context is serialised by the caller. */
int dfk_paths_build(dfk_ctx* ctx, const uint8_t* packed_bases, const uint64_t* base_off, const uint32_t* read_len,
              const uint8_t* pq_bytes, const uint64_t* pq_off, uint64_t n_reads);
int dfk_paths_build_device(dfk_ctx* ctx, const void* d_packed_bases, uint64_t packed_bytes, const void* d_base_off,
              const void* d_read_len, const void* d_pq_bytes, uint64_t pq_nbytes, const void* d_pq_off, uint64_t n_reads);
int dfk_paths_stats(dfk_ctx* ctx, uint64_t* n_reads, uint64_t* n_placed, uint64_t* n_path_edges);
int dfk_paths_write(dfk_ctx* ctx, const char* path);
int dfk_paths_fetch(dfk_ctx* ctx, int32_t* offsets, uint64_t* first_edge, int32_t* edges, uint64_t edges_cap);

/* ---- the two steps DF takes right after StageBuildGraph (10X/DF.cc:550,560), on the paths dfk_paths_build made ----
 * dfk_paths_index_write  writePathsIndex (10X/PathsIndex.cc:23-146): dir/a.paths.inv -- per HBV edge the reads whose path holds
 *                        it, ascending (a stable radix sort of the (edge, read) pairs on the device; a feudal file of unsigned
 *                        long lists) -- and dir/a.countsb (reads per edge, an edge and its involution summed).  Defined for any
 *                        graph; the reference itself overruns below ~870 edges (caveat 1).
 * dfk_dups_write         MarkDups (10X/SecretOps.cc:410-566): a.dup, one byte per PAIR -- reads with the same first edge, offset
 *                        and first five bases of their mate are duplicates; the one whose pair has the highest quality sum (the
 *                        lowest read id among equals) stays.  Grouped in a hash table on the device instead of sorted. */
/* The index is built one range of edges at a time when the room is short or the paths hold 2^31 entries or more (the
 * reference's 30 chunk files, PathsIndex.cc:32-99, are the same idea): no limit on the number of path entries.  Limits, refused
 * loudly (DFK_E_ARG) rather than wrapped: fewer than 2^32 reads on any ONE edge, fewer than 2^29 HBV edges and offsets within
 * +-2^24 (the duplicate key).  dir / path NULL: everything but the files (dfk_paths_digest). */
int dfk_paths_index_write(dfk_ctx* ctx, const char* dir);
int dfk_dups_write(dfk_ctx* ctx, const char* path, uint64_t* n_marked_pairs);
/* Both steps, as DF takes them one after the other (10X/DF.cc:550,560): the same files, with the lists of a.paths.inv leaving
 * the device and entering the file while the duplicates are marked (when the index is built in one range; otherwise exactly the
 * two calls above). */
int dfk_paths_index_dups_write(dfk_ctx* ctx, const char* dir, const char* dup_path, uint64_t* n_marked_pairs);

/* ---- rows f-1 / f-2 / f-4 checked where no oracle runs (tests/test_gpu_fullsize_graph.py; DF prints the digests) ----
 * Replaces nothing in the reference (its nearest thing is hbv.CheckSum() / Validate(hbv, paths), 10X/DF.cc:597-598).
 * dfk_paths_digest  fills out[DFK_CHECK_WORDS] with digests that depend on the CONTENT of a.paths, a.paths.inv, a.countsb and
 *                   a.dup only -- not on the batches the reads were pathed in, the passes the dictionary was counted in or the
 *                   entry numbering the k-mer index holds -- and with the graph's identities:
 *   DFK_CK_PATHS_SUM/XOR    over reads r of h(r, offset, lastSkip, edge ids...), the element as a.paths holds it (after dfk_paths_build)
 *   DFK_CK_N_READS / N_PLACED / N_PATH_EDGES
 *   DFK_CK_INV_SUM/XOR      over positions i of a.paths.inv's data of h(i, read id); DFK_CK_INV_STARTS over edges e of h(e, first
 *                           position of e's list); DFK_CK_INV_ENTRIES = the lists' total length     (after dfk_paths_index_write)
 *   DFK_CK_COUNTSB_DIGEST   over edges of h(e, a.countsb[e]); DFK_CK_COUNTSB_SUM their plain sum; DFK_CK_SELF_INVERSE = entries on
 *                           edges that are their own involution: COUNTSB_SUM = 2 * INV_ENTRIES - SELF_INVERSE
 *   DFK_CK_DUP_DIGEST       over pairs p of h(p, a.dup[p]); DFK_CK_DUP_MARKED the number marked               (after dfk_dups_write)
 *   DFK_CK_EDGE_KMERS       sum of the canonical edges' k-mers (= DFK_CK_N_SOLID: every solid k-mer lies on exactly one edge);
 *   DFK_CK_INV_VIOLATIONS   HBV edges e with inv(inv(e)) != e or inv(e) out of range; DFK_CK_N_EDGES
 *   DFK_CK_VALID            DFK_CK_HAS_* bits: which of the three groups are filled
 *   dfk_paths_index_write(ctx, NULL) / dfk_dups_write(ctx, NULL, &n) do everything but write the files.
 * dfk_paths_verify  a second look at every placed read by a kernel that shares no code with the pather (k_path_verify,
 *                   csrc/dfk_check_kernels.h): out[8] = reads placed; paths with an edge id out of range, two consecutive edges
 *                   that do not meet at a vertex (pathPartsToReadPath cuts there, BuildReadQGraph48.cc:1365-1402) or an offset
 *                   behind the first edge; k-mers of placed reads found in the dictionary at a position the path covers; those
 *                   whose (edge, position) is what the path implies there; placed reads with NO such k-mer (the seed a path was
 *                   built from always is one); placed reads all of whose hits agree; dictionary entries whose edge bases are not
 *                   the k-mer; hits at positions outside the path.  [1], [4], [6] must be 0.  Before dfk_paths_index_write /
 *                   dfk_dups_write (they give the k-mer index back).  The reads are those given to dfk_paths_build*. */
#define DFK_CHECK_WORDS 20
enum { DFK_CK_PATHS_SUM = 0, DFK_CK_PATHS_XOR = 1, DFK_CK_N_READS = 2, DFK_CK_N_PLACED = 3, DFK_CK_N_PATH_EDGES = 4,
       DFK_CK_INV_SUM = 5, DFK_CK_INV_XOR = 6, DFK_CK_INV_STARTS = 7, DFK_CK_INV_ENTRIES = 8,
       DFK_CK_COUNTSB_DIGEST = 9, DFK_CK_COUNTSB_SUM = 10, DFK_CK_SELF_INVERSE = 11,
       DFK_CK_DUP_DIGEST = 12, DFK_CK_DUP_MARKED = 13,
       DFK_CK_EDGE_KMERS = 14, DFK_CK_N_SOLID = 15, DFK_CK_INV_VIOLATIONS = 16, DFK_CK_N_EDGES = 17, DFK_CK_VALID = 18 };
#define DFK_CK_HAS_PATHS 1u
#define DFK_CK_HAS_INDEX 2u
#define DFK_CK_HAS_DUPS  4u
int dfk_paths_digest(dfk_ctx* ctx, uint64_t* out /* [DFK_CHECK_WORDS] */);
int dfk_paths_verify(dfk_ctx* ctx, const uint8_t* packed_bases, const uint64_t* base_off, const uint32_t* read_len, uint64_t n_reads,
              uint64_t* out /* [8] */);
int dfk_paths_verify_device(dfk_ctx* ctx, const void* d_packed_bases, uint64_t packed_bytes, const void* d_base_off, const void* d_read_len,
              uint64_t n_reads, uint64_t* out /* [8] */);

/* ---- the data-parallel half of ParseBarcodedFastqs (10X/ParseBarcodedFastqs.cc:306-539) ----
 * The host inflates the two .gz files, cuts them into lines and decides where every barcode stands in the output (the buckets
 * are the iteration order of a std::unordered_set, :311-336: a container's business).  For one group of pairs it holds, the
 * device then does the rest:
 *   the order of the pairs   by the place of their barcode, inside a barcode DEScending by (read 1, read 2) as base sequences,
 *                            equal pairs in file order (the list insertion of :434-449) -- a stable radix sort on the sequences
 *   bases -> 2-bit codes     N -> A (:407-412), LSB-first as BaseVec stores them (feudal/FieldVec.h:766-770)
 *   qualities -> PQVec       PQVecEncoder::init/encode (feudal/PQVec.cc:18-127), the block choice AND its cut-back rule
 * dfk_pbf_run takes the reads as the fastq has them (ASCII) and returns the group's share of OUT_HEAD.fastb / .qualp ready to be
 * appended: the pairs' order, the reads' lengths, both files' variable data with their offsets.  Errors as the reference's:
 * DFK_E_INPUT for a quality above 63 ("Your input reads are funny", PQVec.cc:30-35) or a character that is no base.
 * Any context will do (K plays no part); its HBM budget bounds the group. */
#define DFK_PBF_DROP 0xFFFFFFFFu
typedef struct dfk_pbf_input {
    const uint8_t*  seq[2];      /* file 1 / file 2: the held reads' base characters, one read behind the other */
    const uint8_t*  qual[2];     /* their quality characters (phred + 33), same layout */
    const uint64_t* off[2];      /* [m + 1]: where read i starts in seq / qual */
    const uint32_t* rank;        /* [m]: the place of pair i's barcode in the output; 0 = unbarcoded (file order kept);
                                    DFK_PBF_DROP = the pair is left out (READS_PER_BC, :328) */
    uint64_t m;                  /* pairs held */
} dfk_pbf_input;
typedef struct dfk_pbf_output {
    uint64_t n_pairs;            /* pairs written: 2 * n_pairs reads, read 2k = read 1 of pair order[k], 2k+1 its mate */
    const uint32_t* order;       /* [n_pairs] */
    const uint32_t* read_len;    /* [2 n_pairs]: the .fastb fixed data */
    const uint8_t*  fastb_var;  const uint64_t* fastb_off;   /* [2 n_pairs + 1], relative to fastb_var */
    const uint8_t*  qualp_var;  const uint64_t* qualp_off;
    float ms_sort, ms_encode, ms_total;   /* device time: the order; packing + PQVec; the whole call with its transfers */
} dfk_pbf_output;
typedef struct dfk_pbf dfk_pbf;
int  dfk_pbf_run(dfk_ctx* ctx, const dfk_pbf_input* in, dfk_pbf** out);
int  dfk_pbf_result(const dfk_pbf* r, dfk_pbf_output* o);      /* pointers stay valid until dfk_pbf_free */
void dfk_pbf_free(dfk_pbf* r);

/* ---- multi-GPU pieces (one process per GPU; the caller owns the RCCL exchange) ----
 * The reference's only exchange is MapReduceEngine's thread all-to-all ("swizzle",
 * MapReduceEngine.h:345-388): every key goes to the thread that owns hash % T.  Here every
 * k-mer goes to the rank that owns its minimizer bucket (owner = bucket & (world-1); world is
 * a power of two), travelling inside 32-byte super-k-mer records:
 *
 *   dfk_shard_begin       trim this rank's reads; returns its k-mer instance count
 *   <caller: all-reduce the instance counts so that every rank derives the same bucket count>
 *   dfk_shard_plan        passes (equal bucket ranges) this rank needs to fit its HBM (caller takes the max over ranks)
 *   for pass in 0 .. 2^log2_passes - 1:
 *     dfk_shard_partition   kmerize the pass's minimizer buckets into records grouped by destination rank
 *     <caller: all-to-all of send_counts; dfk_shard_recv_buffer; all-to-all of the record bytes, over RCCL/xGMI>
 *     dfk_shard_count       regroup the received records by fine bucket and count them; after the
 *                           last pass this rank holds the solid k-mers it owns, pre-adjacency
 *   dfk_shard_adj_queries neighbour keys of the local solid k-mers, grouped by owner rank
 *   <caller: all-to-all of the 16-byte keys>
 *   dfk_shard_adj_answer  presence of each received key in the local solid set
 *   <caller: all-to-all of the answer bytes back>
 *   dfk_shard_adj_apply   clear the context bits whose neighbour is not solid anywhere
 *
 * Afterwards dfk_solid_count/fetch, dfk_spectrum, dfk_good_lens return this rank's share
 * (solid sets are disjoint; spectra add).  superplus_amd/dist.py drives these with
 * torch.distributed; INTEGRATION.md shows the plain RCCL calls. */
int dfk_shard_begin(dfk_ctx* ctx,
              const void* d_packed_bases, uint64_t packed_bytes, const void* d_base_off,
              const void* d_read_len, const void* d_pq_bytes, uint64_t pq_nbytes,
              const void* d_pq_off, const void* d_bc, uint64_t n_reads,
              int64_t global_read_offset /* index of this shard's first read (ign_bc_below is global) */,
              uint64_t* n_inst_local);
/* The same from host memory, for a host that maps the read files (DF NUM_GPUS=N): the tables are this rank's SLICE of
 * the whole set's tables -- base_off / pq_off keep the whole set's offsets and packed_bases / pq_bytes are the whole
 * set's arrays -- and only the bytes of this rank's reads are uploaded.  The device copies live until the next run. */
int dfk_shard_begin_host(dfk_ctx* ctx,
              const uint8_t* packed_bases, const uint64_t* base_off /* n+1 */, const uint32_t* read_len,
              const uint8_t* pq_bytes, const uint64_t* pq_off /* n+1 */, const int32_t* bc, uint64_t n_reads,
              int64_t global_read_offset, uint64_t* n_inst_local);
int dfk_shard_plan(dfk_ctx* ctx, uint32_t world, uint64_t n_inst_global, uint32_t* log2_passes);
int dfk_shard_partition(dfk_ctx* ctx, uint32_t world, uint64_t n_inst_global, uint32_t log2_passes, uint32_t pass,
              const void** d_records, uint64_t* send_counts /* [world], in 32-byte records */);
/* The same in two steps, for a driver that hides the partition of pass p+2 under the count of pass p (while the
 * records of pass p+1 travel): _begin returns the send buffer and the send counts at once -- the counts are known from
 * the counting scan -- and starts the kernels now (defer = 0) or, on the library's second stream, right behind the
 * k_count of the dfk_shard_count call that follows (defer = 1); _end waits for them and checks that every slice
 * received what the scan counted.  The buffer may be sent only after _end.  One partition at a time; the send buffers
 * of two consecutive passes are alive together, the one of pass p is given back when pass p is counted or pass p+2
 * begun.  dfk_shard_partition = _begin(defer = 0) + _end. */
int dfk_shard_partition_begin(dfk_ctx* ctx, uint32_t world, uint64_t n_inst_global, uint32_t log2_passes, uint32_t pass, int defer,
              const void** d_records, uint64_t* send_counts /* [world], in 32-byte records */);
int dfk_shard_partition_end(dfk_ctx* ctx, uint32_t pass);
/* Room for the records this rank is about to receive, from the library's own HBM budget (so that the
 * send, receive and regroup buffers of a pass are all planned in one place); dfk_shard_count frees it.
 * Optional: dfk_shard_count accepts any device pointer. */
int dfk_shard_recv_buffer(dfk_ctx* ctx, uint64_t n_records, void** d_buf);
int dfk_shard_count(dfk_ctx* ctx, const void* d_records, uint64_t n_records, uint32_t pass);
int dfk_shard_adj_queries(dfk_ctx* ctx, const void** d_keys,
              uint64_t* send_counts /* [world], in 16-byte keys */);
int dfk_shard_adj_answer(dfk_ctx* ctx, const void* d_keys, uint64_t n_keys, void* d_present /* u8[n_keys] */);
int dfk_shard_adj_apply(dfk_ctx* ctx, const void* d_present, uint64_t n_keys);
/* The whole dictionary on ONE rank, for what the reference does with it after createDict whatever its thread count
 * (buildEdges, buildHBVFromEdges, pathReads: BuildReadQGraph48.cc:1636,1664): every other rank sends the entries
 * dfk_shard_dict_share names (32 bytes each, contexts final) to the gathering rank, which receives them into the room
 * dfk_shard_dict_adopt returns (it first gives back what the run's exchange still held: this rank's staged reads, the
 * send / receive buffers) and then declares the dictionary whole: from dfk_shard_dict_whole on the context answers as
 * after a single-GPU count (dfk_solid_count, dfk_write_kvec, dfk_graph_build, dfk_paths_build); the spectrum stays this
 * rank's share. */
int dfk_shard_dict_share(dfk_ctx* ctx, const void** d_entries, uint64_t* n_entries);
int dfk_shard_dict_adopt(dfk_ctx* ctx, uint64_t n_entries, void** d_room);
int dfk_shard_dict_whole(dfk_ctx* ctx);

/* ---- rows f-2 / f-4 of a multi-GPU run: every rank holds the whole dictionary (dfk_shard_dict_adopt on every rank) and has built
 * the same graph; the reads stay sharded by pair range as for the count.  The reference's pathReads is a parallelFor over the reads
 * (paths/long/BuildReadQGraph48.cc:1420-1442), writePathsIndex a sort of (edge, read) pairs (10X/PathsIndex.cc:23-146), MarkDups a
 * group-by (10X/SecretOps.cc:410-566); the caller owns the exchanges (csrc/df_shard.h):
 *   dfk_paths_build(ctx, NULL...)   paths the pair range dfk_shard_begin_host staged (kept under DFK_F_KEEP_INPUTS)
 *   dfk_paths_var_bytes             this rank's share of a.paths' variable data  <caller: all-gather>
 *   dfk_paths_write_part            its elements and offset-table entries at their place in the file (rank of read 0: the control block)
 *   dfk_shard_pidx_pairs            its (edge << 32 | whole-set read id) pairs sorted by edge; send_counts[r] = pairs on the edges
 *                                   rank r owns (edges [r n / world, (r+1) n / world)); counts_host[n_edges] = its reads per edge
 *   <caller: all-to-all of the 8-byte pairs; sum of counts_host over the ranks>
 *   dfk_shard_pidx_write            merges what arrived (stable by edge: sources hold ascending read ranges) and writes this rank's
 *                                   range of a.paths.inv; rank 0 also a.countsb
 *   dfk_shard_dup_keys              {key, score} of every placed read grouped by owner rank = hash(key) % world (16 bytes each)
 *   <caller: all-to-all>            dfk_shard_dup_answer: a byte per received item (1 = not its group's best)  <caller: back>
 *   dfk_shard_dup_write             marks this rank's pairs and writes their bytes of a.dup (rank of pair 0: the header)
 * dfk_paths_digest then returns THIS rank's share of the digests: sums add and xors xor over the ranks. */
int dfk_paths_var_bytes(dfk_ctx* ctx, uint64_t* bytes);
int dfk_paths_write_part(dfk_ctx* ctx, const char* path, uint64_t first_read, uint64_t total_reads, uint64_t var_before, uint64_t var_total);
int dfk_shard_pidx_pairs(dfk_ctx* ctx, uint32_t world, const void** d_pairs, uint64_t* send_counts /* [world] */, uint64_t* counts_host /* [n_edges] */);
int dfk_shard_pidx_write(dfk_ctx* ctx, uint32_t world, uint32_t rank, const void* d_pairs_in, uint64_t n_in, const uint64_t* counts_global /* [n_edges] */, const char* dir);
int dfk_shard_dup_keys(dfk_ctx* ctx, uint32_t world, const void** d_items, uint64_t* send_counts /* [world], 16-byte items */);
int dfk_shard_dup_answer(dfk_ctx* ctx, const void* d_items_in, uint64_t n_in, void* d_answers /* u8[n_in] */);
int dfk_shard_dup_write(dfk_ctx* ctx, const void* d_answers_back, uint64_t n, const char* path, uint64_t first_pair, uint64_t total_pairs, uint64_t* n_marked);

#ifdef __cplusplus
}
#endif
#endif
