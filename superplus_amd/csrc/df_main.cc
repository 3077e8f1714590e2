// superplus_amd/csrc/df_main.cc -- `DF` front-end for seam B2: keeps the reference's
// command line (`DF ROOT=... LR=a.fastb[,b.fastb] [K=48 MIN_FREQ=3 MIN_BC=2 MIN_QUAL=7 OUT_DIR= ...]`,
// 10X/DF.cc:79-195) and its on-disk inputs and outputs for the ingest + createDict part of the run:
//
//   reads  LR heads' .fastb/.qualp/.bci                                      (DF.cc:251-258)
//   writes work_dir/data/frag_reads_orig.{fastb,qualp,bci}   LoadData        (10X/DfTools.cc:69-170)
//          work_dir/data/frag_reads_orig.{lens,qhist,dti}    FirstLoadData   (DF.cc:50-68)
//          work_dir/subsam.{names,starts}                                    (DF.cc:263-265,477-482)
//          work_dir/stats/histogram_kmer_count.json          WriteKmerSpectrum (BuildReadQGraph48.cc:283-285)
//          work_dir/kmers.kvec                               the dictionary  (BuildReadQGraph48.cc:287-288)
//          work_dir/data/frag_reads_orig.1000.{fastb,qualp}  WriteSubSample  (10X/DfTools.cc:32-67)
//          work_dir/a.<K>/{a.k,a.hbv,a.hbx,a.to_left,a.to_right,a.inv,a.fastb,a.kmers}  WriteAssemblyFiles (10X/WriteFiles.cc:69-101)
//
// The hot path itself runs in libdfk (HIP); this file is host plumbing only -- but plumbing for 1.8 G reads:
// the inputs are mapped, not copied; the one-input LR_SELECT_FRAC=1 case (what runall.sh:127 runs) re-emits
// the read files as parallel range copies while the GPU counts; lens, the quality histogram and the barcode
// expansion run on NUM_THREADS host threads.  Not reproduced: everything after createDict (graph build,
// pathing, the other seven stages), which are out of scope.
//
// Arguments beyond the reference's: HBM_GB= (device memory the library may use; 0 = 90 % of what is free --
// MAX_MEM_GB keeps its reference meaning, a HOST memory cap (system/System.cc:1073-1078), and is not a device
// budget), DEVICE=, KVEC= / KVEC_SORTED= (write kmers.kvec -- by default only when GRAPH=False; in ascending k-mer order), GRAPH= (build the graph and
// write a.<K>/: on by default, as in the reference), MINIMIZER=.
#include "../../include/dfk.h"
#include "df_shard.h"
#include "feudal_io.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <functional>
#include <exception>
#include <cerrno>
#include <map>
#include <mutex>
#include <memory>
#include <sstream>
#include <string>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <signal.h>
#include <sys/wait.h>
#include <thread>
#include <type_traits>
#include <unistd.h>

namespace {

std::string date()
{ time_t t = time(nullptr); char b[64]; strftime(b, sizeof b, "%a %b %d %H:%M:%S %Y", localtime(&t)); return b; }

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void mkpath(const std::string& p)
{
    for (size_t i = 1; i <= p.size(); ++i)
        if (i == p.size() || p[i] == '/') mkdir(p.substr(0, i).c_str(), 0777);
}

bool is_file(const std::string& p) { struct stat s; return stat(p.c_str(), &s) == 0 && S_ISREG(s.st_mode); }

[[noreturn]] void give_up(const std::string& msg) { printf("\n%s\nGiving up.\n", msg.c_str()); exit(1); }

bool truthy(const std::string& v) { return v == "True" || v == "true" || v == "1" || v == "yes"; }

// "{a,b}" or "a,b" -> list (ParseStringSet on "{" + LR + "}", DF.cc:247-248)
std::vector<std::string> parse_set(std::string s)
{
    s.erase(std::remove(s.begin(), s.end(), '{'), s.end());
    s.erase(std::remove(s.begin(), s.end(), '}'), s.end());
    std::vector<std::string> out; std::stringstream ss(s); std::string tok;
    while (std::getline(ss, tok, ',')) if (!tok.empty()) out.push_back(tok);
    return out;
}

// The reference's global random stream (random/RNGen.h:28-84, RNGen.cc:17-18): additive lagged Fibonacci over 31
// words seeded from 1 by x -> x*1103515245 + 12345, front at word 3, rear at word 0, 310 draws discarded; a draw is
// the new front word >> 1.  (The reference keeps the words in unsigned long; only their low 32 bits ever matter.)
struct RefRandom {
    uint32_t st[31]; int f = 3, r = 0;
    RefRandom()
    { uint32_t last = 1; st[0] = last; for (int i = 1; i < 31; ++i) st[i] = last = last * 1103515245u + 12345u; for (int n = 0; n < 310; ++n) next(); }
    long next()
    {
        const uint32_t result = (st[f] += st[r]);
        if (++f >= 31) { f = 0; ++r; }
        else if (++r >= 31) r = 0;
        return (long)(result >> 1);
    }
};

struct DataSet { uint8_t dt; uint8_t pad[7]; int64_t start; };   // 10X/DfTools.h:23-45; dt 2 = UNBAR_10X, 3 = BAR_10X
static_assert(sizeof(DataSet) == 16, "DataSet is 16 bytes");

unsigned g_threads = 1;

// fn(t, lo, hi) over [0, n) cut into one contiguous range per thread
unsigned g_side_threads = 0;     // (0 = all) what the side-file passes may use while the upload's lanes want the cores
void parallel_ranges(uint64_t n, const std::function<void(unsigned, uint64_t, uint64_t)>& fn, uint64_t min_per_thread = 1 << 16, bool side = false)
{
    const unsigned most = side && g_side_threads ? std::min(g_side_threads, g_threads) : g_threads;
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(most, n / std::max<uint64_t>(1, min_per_thread)));
    if (T <= 1) { fn(0, 0, n); return; }
    std::vector<std::thread> th;
    std::exception_ptr err; std::atomic<bool> failed{false};
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back([&, t] { try { fn(t, n * t / T, n * (t + 1) / T); } catch (...) { if (!failed.exchange(true)) err = std::current_exception(); } });
    for (auto& x : th) x.join();
    if (failed) std::rethrow_exception(err);
}

// unaligned little-endian loads (tables inside a mapped feudal file sit wherever the var data ends)
inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

struct Mapped {
    const uint8_t* p = nullptr; size_t n = 0; int fd = -1;
    void open(const std::string& path)
    {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        struct stat s; if (fstat(fd, &s) != 0) throw std::runtime_error("cannot stat " + path);
        n = (size_t)s.st_size;
        if (n) {
            void* m = mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
            if (m == MAP_FAILED) throw std::runtime_error("cannot map " + path);
            p = (const uint8_t*)m;
            // Read once, front to back -- and, what matters here, VM_SEQ_READ: when a mapping's pages leave the page table (munmap, or
            // the MADV_DONTNEED behind the upload's copies) the kernel marks every page that was touched as accessed, which moves
            // it to the active list under one lock -- unless the mapping is sequential.  On a freshly written tmpfs file that was
            // the whole difference between 25 and 130 GB/s for the first reading (tools/fs_read_after_write.cc), and the three
            // seconds the unmapping used to take.
            (void)madvise(m, n, MADV_SEQUENTIAL);
        }
    }
    ~Mapped() { unmap(); if (fd >= 0) close(fd); }
    void unmap()                                                        // the descriptor stays open
    {
        if (!p) return;
        const auto t0 = std::chrono::steady_clock::now();
        // from the end, a piece at a time: an unmap of tens of GB holds the address-space lock for seconds, and other threads
        // (the GPU runtime pinning a buffer for a copy, say) wait for it that long; a piece is some tens of milliseconds
        if (p) {
            const size_t piece = (size_t)256 << 20, page = 4096;
            size_t end = (n + page - 1) / page * page;
            while (end > 0) { const size_t lo = end > piece ? (end - piece) / page * page : 0; munmap((void*)(p + lo), end - lo); end = lo; }
        }
        if (getenv("DFK_TRACE")) fprintf(stderr, "[DF] unmapped %.2f GB in %.3f s\n", n / 1e9, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        p = nullptr;
    }
};

// a mapped feudal file: control block, var data, absolute offset table, fixed data (feudal_io.h has the layout)
struct FeudalMap {
    Mapped m; std::string path; uint64_t n = 0, varTab = 0, fixedOff = 0;
    void open(const std::string& pth, bool check_table = true)
    {
        path = pth; m.open(pth);
        if (m.n < 24) throw std::runtime_error(path + ": too short for a feudal file");
        feudal::Header h; memcpy(&h, m.p, 24);
        if ((h.flags & 3) != 1) throw std::runtime_error(path + ": not a single-file feudal file");
        if (h.varTab < 24 || h.fixedOff < h.varTab || h.fixedOff > m.n || (h.fixedOff - h.varTab) % 8 || h.fixedOff == h.varTab)
            throw std::runtime_error(path + ": inconsistent feudal control block");
        n = (h.fixedOff - h.varTab) / 8 - 1; varTab = h.varTab; fixedOff = h.fixedOff;
        if ((uint32_t)n != h.n) throw std::runtime_error(path + ": element count mismatch");
        if (check_table) check_offsets();
    }
    // offsets: absolute, inside the var data, ascending.  `sizes` (optional): element i must take exactly sizes(i) bytes -- the
    // .fastb check that a read's bytes are ceil(len/4), in the same pass.
    struct NoSizes { static constexpr bool given = false; uint64_t operator()(uint64_t) const { return 0; } };
    void check_offsets() const { check_offsets(NoSizes{}, nullptr); }
    template <class Sizes>                                               // (a functor the loop inlines: it runs 1.8e9 times)
    void check_offsets(const Sizes& sizes, const char* what_else) const
    {
        constexpr bool with_sizes = !std::is_same<Sizes, NoSizes>::value;
        std::atomic<int> bad{0};
        // (entering a range's pages into the page table in one call first -- MADV_POPULATE_READ -- made the pass SLOWER here: 0.45 s
        // against 0.29 at configs[1])
        parallel_ranges(n + 1, [&](unsigned, uint64_t lo, uint64_t hi) {
            uint64_t prev = lo ? off(lo - 1) : 24;
            for (uint64_t i = lo; i < hi; ++i) {
                const uint64_t o = off(i);
                if (o < prev || o > varTab) { bad = 1; return; }
                if (with_sizes && i && o - prev != sizes(i - 1)) { bad = 2; return; }
                prev = o;
            }
        });
        if (bad == 1) throw std::runtime_error(path + ": offset table is not ascending or runs past the data");
        if (bad == 2) throw std::runtime_error(path + (what_else ? what_else : ": an element has not the size it should"));
    }
    uint64_t off(uint64_t i) const { return ld64(m.p + varTab + 8 * i); }       // absolute file offset of element i
    const uint8_t* off_table() const { return m.p + varTab; }
    const uint8_t* fixed() const { return m.p + fixedOff; }
};

// pwrite the whole range, from several threads
void write_range(int fd, const uint8_t* src, uint64_t bytes, uint64_t file_off)
{
    constexpr uint64_t PIECE = 8ull << 20;
    const uint64_t n_pieces = (bytes + PIECE - 1) / PIECE;
    std::atomic<uint64_t> next{0};
    std::atomic<bool> bad{false};
    auto work = [&] {
        for (uint64_t i; (i = next.fetch_add(1)) < n_pieces && !bad;) {
            const uint64_t o = i * PIECE, len = std::min(PIECE, bytes - o);
            for (uint64_t done = 0; done < len;) {
                const ssize_t w = pwrite(fd, src + o + done, len - done, (off_t)(file_off + o + done));
                if (w <= 0) { bad = true; return; }
                done += (uint64_t)w;
            }
        }
    };
    // (one writer per file: threads writing ONE file serialise on its page-cache lock -- measured on the GPU box's tmpfs:
    // 8.6 GB/s from one thread, 6.6 GB/s from sixteen; tools/fs_write_scaling.cc)
    const unsigned T = 1;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    if (bad) throw std::runtime_error("short write");
}

// A feudal file re-emitted from its mapped image: our control block (the reference's writer constants), then the
// var data, offset table and fixed data as they are.
// LINK_READS=True: when the input already is, byte for byte, the file that would be written (it carries the control block
// the reference's writer gives such a file, and LoadData keeps every read in place), frag_reads_orig.* can be another name
// for the same data instead of a second copy of tens of GB -- a hard link; anything in the way (another file system, a
// different control block) falls back to the copy.
bool link_feudal(const FeudalMap& in, const std::string& path, uint8_t szFixed, uint8_t szX, uint8_t szA)
{
    feudal::Header h{(uint32_t)in.n, 1, szFixed, szX, szA, in.varTab, in.fixedOff};
    if (memcmp(&h, in.m.p, 24) != 0) return false;
    unlink(path.c_str());
    return link(in.path.c_str(), path.c_str()) == 0;
}

void copy_feudal(const FeudalMap& in, const std::string& path, uint8_t szFixed, uint8_t szX, uint8_t szA)
{
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) throw std::runtime_error("cannot create " + path);
    feudal::Header h{(uint32_t)in.n, 1, szFixed, szX, szA, in.varTab, in.fixedOff};
    bool ok = pwrite(fd, &h, 24, 0) == 24;
    if (ok) { try { write_range(fd, in.m.p + 24, in.m.n - 24, 24); } catch (...) { ok = false; } }
    ok = (close(fd) == 0) && ok;
    if (!ok) throw std::runtime_error("short write " + path);
}

// GetQualStats (10X/DfTools.cc:172-238): hist[parity][pos][q] over all reads, q in 0..255 while counting.  A block
// of equal quals (nBits = 0: every constant-quality stretch) is booked as a +1/-1 pair on a per-(parity,q) difference
// row and integrated at the end; other blocks base by base.
struct QualHist {
    int max_len; std::vector<int64_t> direct, diff;   // [2][max_len][256], [2][256][max_len + 1]
    explicit QualHist(int ml) : max_len(ml), direct((size_t)2 * ml * 256, 0), diff((size_t)2 * 256 * (ml + 1), 0) {}
    void add_read(uint64_t r, const uint8_t* p, const uint8_t* end)
    {
        const size_t par = r & 1;
        int pos = 0;
        while (p < end && *p) {                                            // PQVec blocks (feudal/PQVec.cc:87-127)
            if (p + 3 > end) break;
            const unsigned nQs = p[0], hdr = p[1] | (p[2] << 8), nBits = hdr & 7, minQ = (hdr >> 3) & 63;
            const uint64_t blk = ((uint64_t)nQs * nBits + 24) >> 3;
            if (p + blk > end) break;
            if (nBits == 0) {
                const int a = std::min(pos, max_len), b = std::min(pos + (int)nQs, max_len);
                int64_t* row = &diff[(par * 256 + minQ) * (size_t)(max_len + 1)];
                row[a]++; row[b]--;
                pos += (int)nQs;
            } else {
                uint64_t bit = 17;
                for (unsigned i = 0; i < nQs; ++i, bit += nBits) {
                    const unsigned w = p[bit >> 3] | ((bit >> 3) + 1 < blk ? (unsigned)p[(bit >> 3) + 1] << 8 : 0u);
                    const unsigned q = minQ + ((w >> (bit & 7)) & ((1u << nBits) - 1));
                    if (pos < max_len) direct[(par * max_len + pos) * 256 + q]++;
                    ++pos;
                }
            }
            p += blk;
        }
    }
    void merge_into(std::vector<int64_t>& total) const            // total: [2][max_len][256]
    {
        for (size_t i = 0; i < direct.size(); ++i) total[i] += direct[i];
        for (size_t par = 0; par < 2; ++par)
            for (size_t q = 0; q < 256; ++q) {
                const int64_t* row = &diff[(par * 256 + q) * (size_t)(max_len + 1)];
                int64_t run = 0;
                for (int pos = 0; pos < max_len; ++pos) { run += row[pos]; total[(par * max_len + pos) * 256 + q] += run; }
            }
    }
};

struct Timing { double read = 0, ingest_out = 0, upload = 0, count = 0, fetch_write = 0, total = 0; };

// WriteHistToJson<int64_t> as WriteKmerSpectrum calls it (10X/MakeHist.cc:67-92): the text dfk_spectrum_json produces,
// here for a spectrum summed over ranks
std::string spectrum_json(const std::vector<uint64_t>& hist)
{
    std::string s = "{\n\t\"description\": \"kmer_count\",\n\t\"stage\": \"DF\",\n\t\"binsize\": 1,\n\t\"min\": 0,\n";
    s += "\t\"max\": " + std::to_string((long long)hist.size() - 1) + ",\n";
    s += "\t\"numbins\": " + std::to_string(hist.size()) + ",\n\t\"vals\": [";
    for (size_t i = 0; i < hist.size(); ++i) { s += std::to_string(hist[i]); if (i + 1 != hist.size()) s += ","; }
    return s + "]\n}\n";
}

// One rank of `DF NUM_GPUS=N`: its pair range of the (single, already LoadData-ordered) input, the sharded createDict
// over the transport, its share of kmers.kvec; rank 0 writes the spectrum summed over the ranks.
// This process's threads on the CPUs of the GPU's NUMA node: the transfer lanes copy between the page cache, their pinned
// buffers and the device, and from the other socket that is half the rate (measured on a two-socket box: the whole stage
// 20.0 -> 18.6 s).  Threads started afterwards inherit it.  DF_NO_BIND=1 leaves the threads where the scheduler puts them.
void bind_to_gpu_node(int device)
{
    if (getenv("DF_NO_BIND")) return;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (char* p = bus; *p; ++p) *p = (char)tolower(*p);
    int node = -1;
    { FILE* f = fopen((std::string("/sys/bus/pci/devices/") + bus + "/numa_node").c_str(), "r"); if (!f) return; if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return;
    char list[4096] = {0};
    { FILE* f = fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r"); if (!f) return; if (!fgets(list, sizeof list, f)) list[0] = 0; fclose(f); }
    cpu_set_t set; CPU_ZERO(&set);
    int n_cpus = 0;
    for (char* p = list; *p && *p != '\n';) {
        char* e; const long a = strtol(p, &e, 10); long b = a;
        if (e == p) break;
        if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &set); ++n_cpus; }
        p = *e == ',' ? e + 1 : e;
    }
    if (n_cpus && sched_setaffinity(0, sizeof set, &set) == 0 && getenv("DFK_TRACE")) fprintf(stderr, "[DF] bound to NUMA node %d (%d CPUs) of device %d\n", node, n_cpus, device);
}

// One machine-readable line with the content digests of a.paths / a.paths.inv / a.countsb / a.dup and the graph's identities
// (dfk_paths_digest): what two runs of a stage this size are compared by (the reference prints hbv.CheckSum() for the same
// purpose, 10X/DF.cc:598).
void print_digests(dfk_ctx* ctx, const uint64_t* whole_run = nullptr)
{
    uint64_t w[DFK_CHECK_WORDS] = {};
    if (whole_run) memcpy(w, whole_run, sizeof w);
    else if (dfk_paths_digest(ctx, w)) return;
    printf("DF_DIGESTS {\"a.paths\": \"%016llx%016llx\", \"a.paths.inv\": \"%016llx%016llx%016llx\", \"a.countsb\": \"%016llx\", \"a.dup\": \"%016llx\", "
           "\"reads\": %llu, \"placed\": %llu, \"path_edges\": %llu, \"index_entries\": %llu, \"countsb_sum\": %llu, \"self_inverse_entries\": %llu, \"dup_pairs\": %llu, "
           "\"edge_kmers\": %llu, \"solid\": %llu, \"involution_violations\": %llu, \"hbv_edges\": %llu}\n",
           (unsigned long long)w[DFK_CK_PATHS_SUM], (unsigned long long)w[DFK_CK_PATHS_XOR], (unsigned long long)w[DFK_CK_INV_SUM], (unsigned long long)w[DFK_CK_INV_XOR],
           (unsigned long long)w[DFK_CK_INV_STARTS], (unsigned long long)w[DFK_CK_COUNTSB_DIGEST], (unsigned long long)w[DFK_CK_DUP_DIGEST],
           (unsigned long long)w[DFK_CK_N_READS], (unsigned long long)w[DFK_CK_N_PLACED], (unsigned long long)w[DFK_CK_N_PATH_EDGES], (unsigned long long)w[DFK_CK_INV_ENTRIES],
           (unsigned long long)w[DFK_CK_COUNTSB_SUM], (unsigned long long)w[DFK_CK_SELF_INVERSE], (unsigned long long)w[DFK_CK_DUP_MARKED],
           (unsigned long long)w[DFK_CK_EDGE_KMERS], (unsigned long long)w[DFK_CK_N_SOLID], (unsigned long long)w[DFK_CK_INV_VIOLATIONS], (unsigned long long)w[DFK_CK_N_EDGES]);
}

int rank_main(std::map<std::string, std::string>& a, unsigned K, const std::string& work_dir, const std::string& head, int rank, int world,
              dfkx::LoopbackHub* hub)
{
    const double t_start = now_s();
    if (const char* hook = getenv("DF_TEST_RANK_FATE")) {
        // test hook (tests/test_df_frontend.py): "die:R" -- rank R ends at once as a crashed process would; every other rank
        // then waits as if inside a collective with no timeout.  What must happen: DF notices and stops them.
        if (!strncmp(hook, "die:", 4)) { if (atoi(hook + 4) == rank) _exit(134); for (;;) pause(); }
    }
    try {
        FeudalMap fb, qp;
        fb.open(head + ".fastb"); qp.open(head + ".qualp");
        const std::vector<int64_t> bci = feudal::read_bci(head + ".bci");
        if (qp.n != fb.n || bci.size() < 2 || bci[0] != 0 || (uint64_t)bci.back() != fb.n || fb.n % 2) throw std::runtime_error(head + ": not a LoadData-ordered pair set");
        // (the ranks read the input by pair range: that is LoadData's order only if every barcode holds whole pairs)
        for (size_t b = 0; b + 1 < bci.size(); ++b) if (bci[b] > bci[b + 1] || (bci[b + 1] - bci[b]) % 2) throw std::runtime_error(head + ".bci: a barcode with an odd number of reads; NUM_GPUS > 1 needs whole pairs per barcode");
        const uint64_t n_pairs = fb.n / 2, lo = 2 * (n_pairs * (uint64_t)rank / (uint64_t)world), hi = 2 * (n_pairs * (uint64_t)(rank + 1) / (uint64_t)world), n = hi - lo;
        std::vector<int32_t> bc(n, 0);                                                       // DF.cc:447-452 for this rank's reads
        parallel_ranges(bci.size() - 1, [&](unsigned, uint64_t b0, uint64_t b1) {
            for (uint64_t b = b0; b < b1; ++b)
                for (int64_t r = std::max<int64_t>(bci[b], (int64_t)lo); r < std::min<int64_t>(bci[b + 1], (int64_t)hi); ++r) bc[r - lo] = (int32_t)b;
        }, 1024);
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device");
        dfk_config cfg{};
        cfg.abi_version = DFK_ABI_VERSION; cfg.K = K; cfg.min_qual = (uint32_t)atoi(a["MIN_QUAL"].c_str());
        cfg.min_freq = (uint32_t)atoi(a["MIN_FREQ"].c_str()); cfg.min_bc = (uint32_t)atoi(a["MIN_BC"].c_str());
        cfg.device = hub ? atoi(a["DEVICE"].c_str()) : rank % ndev; cfg.ign_bc_below = 0;
        if (!hub) bind_to_gpu_node(cfg.device);
        cfg.minimizer_len = (uint32_t)atoi(a["MINIMIZER"].c_str());
        cfg.hbm_budget_bytes = (uint64_t)(atof(a["HBM_GB"].c_str()) * 1073741824.0);
        if (truthy(a["GRAPH"]) && truthy(a["PATHS"])) cfg.flags |= DFK_F_KEEP_INPUTS;      // the rank's staged reads stay for pathReads
        if (hub && !cfg.hbm_budget_bytes) { size_t fr = 0, tot = 0; (void)hipSetDevice(cfg.device); (void)hipMemGetInfo(&fr, &tot); cfg.hbm_budget_bytes = (uint64_t)(0.8 * (double)fr / world); }   // ranks sharing one GPU
        dfk_ctx* ctx = nullptr;
        if (dfk_create(&cfg, &ctx)) throw std::runtime_error(dfk_last_error());
        std::unique_ptr<dfkx::Transport> T;
        if (hub) T.reset(new dfkx::LoopbackTransport(*hub, rank));
        else T.reset(new dfkx::RcclTransport(rank, world, cfg.device, work_dir + "/.dfk_rccl_id"));
        uint64_t n_local = 0;
        double t0 = now_s();
        int rc = dfk_shard_begin_host(ctx, fb.m.p, (const uint64_t*)(fb.off_table() + 8 * lo), (const uint32_t*)(fb.fixed() + 4 * lo), qp.m.p,
                                      (const uint64_t*)(qp.off_table() + 8 * lo), bc.data(), n, (int64_t)lo, &n_local);
        // (a rank whose own data fail still has to meet the others in the first collective: the status travels with the count)
        uint64_t st0[2] = {rc ? (uint64_t)(-rc) : 0, 0};
        const std::string begin_err = rc ? dfk_last_error() : "";
        T->all_reduce(st0, 1, true);
        if (rc) throw dfkx::ShardError(rc, begin_err);
        if (st0[0]) throw dfkx::ShardError(-(int)st0[0], "another rank failed in dfk_shard_begin; this rank stops with it");
        const double t_begin = now_s() - t0;
        dfkx::ShardTimes tm;
        const uint64_t piece = getenv("DFK_A2A_PIECE_BYTES") ? (uint64_t)atoll(getenv("DFK_A2A_PIECE_BYTES")) : (uint64_t)1 << 30;
        dfkx::shard_create_dict(ctx, *T, n_local, piece, &tm);
        // ---- results: spectrum summed over the ranks (rank 0 writes it), every rank its share of kmers.kvec
        t0 = now_s();
        const int64_t* h = nullptr; uint64_t nb = 0;
        dfk_spectrum(ctx, &h, &nb);
        uint64_t nb_all = nb; T->all_reduce(&nb_all, 1, true);
        std::vector<uint64_t> hist(nb_all, 0);
        for (uint64_t i = 0; i < nb; ++i) hist[i] = (uint64_t)h[i];
        if (nb_all) T->all_reduce(hist.data(), (int)nb_all, false);
        if (rank == 0) { const std::string js = spectrum_json(hist); FILE* f = fopen((work_dir + "/stats/histogram_kmer_count.json").c_str(), "wb"); if (!f) throw std::runtime_error("cannot write spectrum"); fwrite(js.data(), 1, js.size(), f); fclose(f); }
        uint64_t mine[2] = {0, n_local}; dfk_solid_count(ctx, &mine[0]);
        std::vector<uint64_t> all(2 * (size_t)world);
        T->all_gather(mine, 2, all.data());
        uint64_t first = 0, total = 0, inst = 0;
        for (int r = 0; r < world; ++r) { if (r < rank) first += all[2 * r]; total += all[2 * r]; inst += all[2 * r + 1]; }
        const bool want_kvec = a["KVEC"] == "Auto" ? !truthy(a["GRAPH"]) : truthy(a["KVEC"]);
        if (want_kvec && dfk_write_kvec_part(ctx, (work_dir + "/kmers.kvec").c_str(), 0, first, total)) throw std::runtime_error(dfk_last_error());
        const double t_write = now_s() - t0;
        uint64_t done = 1; T->all_reduce(&done, 1, false);                                   // every share is in the file
        dfk_stats st{}; dfk_get_stats(ctx, &st);
        // ---- buildEdges / buildHBVFromEdges / pathReads run on the whole dictionary whatever built it (BuildReadQGraph48.cc:
        //      1636,1664): the shares go to rank 0, which builds a.<K>/ as the single-GPU run does
        double t_gather = 0, t_graph = 0, t_paths = 0;
        uint64_t g_e = 0, g_v = 0, p_placed = 0;
        dfkx::ShardPathTimes pt;
        if (truthy(a["GRAPH"])) {
            // Every rank receives the whole dictionary and builds the same graph (1.4 s at configs[1], deterministic); then the
            // stage stays sharded: a rank paths ITS pair range, the paths index and the duplicate marks are one all-to-all each
            // (df_shard.h).  Rank 0 writes the graph's files meanwhile.
            t0 = now_s();
            dfkx::shard_allgather_dict(ctx, *T, piece);
            t_gather = now_s() - t0;
            t0 = now_s();
            if (rank == 0) printf("%s: finding edge sequences.\n", date().c_str());
            int grc = dfk_graph_build(ctx);
            const std::string gerr = grc ? dfk_last_error() : "";
            { uint64_t worst = grc ? (uint64_t)(-grc) : 0; T->all_reduce(&worst, 1, true); if (grc) throw dfkx::ShardError(grc, gerr); if (worst) throw dfkx::ShardError(-(int)worst, "another rank failed building the graph; this rank stops with it"); }
            const std::string dir = work_dir + "/a." + std::to_string(K);
            std::string bg_fail;
            std::thread graph_writer;
            if (rank == 0) { mkpath(dir); graph_writer = std::thread([&] { if (dfk_graph_write(ctx, dir.c_str())) bg_fail = dfk_last_error(); }); }
            struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_graph{graph_writer};
            dfk_graph_stats(ctx, nullptr, &g_v, &g_e);
            { uint64_t made = 1; T->all_reduce(&made, 1, false); }                             // (the directory exists before anybody writes into it)
            t_graph = now_s() - t0;
            if (truthy(a["PATHS"])) {
                t0 = now_s();
                if (rank == 0) printf("%s: pathing reads\n", date().c_str());
                uint64_t words[DFK_CHECK_WORDS] = {};
                dfkx::shard_paths_index_dups(ctx, *T, dir, lo, fb.n, piece, &pt, words);
                p_placed = pt.placed;
                t_paths = now_s() - t0;
                if (rank == 0) {
                    printf("%.2f%% of pairs appear to be duplicates\n", fb.n ? 100.0 * (double)pt.dup_pairs / (double)(fb.n / 2) : 0.0);
                    print_digests(ctx, words);
                }
            }
            if (graph_writer.joinable()) graph_writer.join();
            if (!bg_fail.empty()) throw std::runtime_error(bg_fail);
        }
        T.reset();
        dfk_destroy(ctx);
        if (rank == 0) {
            printf("%s: dictionary covers %llu kmers\n", date().c_str(), (unsigned long long)total);
            printf("DF_TIMING {\"ranks\": %d, \"reads\": %llu, \"kmer_instances\": %llu, \"solid\": %llu, \"rank0\": {\"upload_trim_s\": %.3f, \"plan_s\": %.3f, "
                   "\"partition_s\": %.3f, \"exchange_wait_s\": %.3f, \"count_s\": %.3f, \"adjacency_s\": %.3f, \"create_dict_s\": %.3f, \"spectrum_kvec_write_s\": %.3f, "
                   "\"bytes_sent_to_peers\": %llu, \"passes\": %u, \"gather_dict_s\": %.3f, \"graph_s\": %.3f, \"paths_s\": %.3f, \"path_reads_s\": %.3f, \"paths_write_s\": %.3f, \"paths_index_s\": %.3f, \"mark_dups_s\": %.3f, "
                   "\"graph_edges\": %llu, \"reads_placed\": %llu, \"rank_total_s\": %.3f}}\n", world, (unsigned long long)fb.n, (unsigned long long)inst,
                   (unsigned long long)total, t_begin, tm.plan, tm.partition, tm.exchange_wait, tm.count, tm.adjacency, tm.total, t_write,
                   (unsigned long long)tm.bytes_sent, tm.n_passes, t_gather, t_graph, t_paths, pt.paths, pt.paths_write, pt.index, pt.dups, (unsigned long long)g_e, (unsigned long long)p_placed, now_s() - t_start);
        }
        return 0;
    } catch (const dfkx::ShardError& e) {
        if (e.code == DFK_E_NOGOOD) { if (rank == 0) printf("\nLooks like your input data have almost no good bases.\nGiving up.\n\n"); return 1; }
        fprintf(stderr, "DF[rank %d]: %s\n", rank, e.what());
        return e.code == DFK_E_NOMEM ? 185 : 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "DF[rank %d]: %s\n", rank, e.what());
        return 1;
    }
}

// The rank processes of `DF NUM_GPUS=N`, watched from the moment they exist.  A rank that dies (a GPU fault, the OOM
// killer, a failed ncclCommInitRank) leaves its peers inside a collective that has no timeout: the first child that exits
// badly takes the others with it -- SIGTERM, then SIGKILL after a grace period -- and DF exits non-zero.
struct ChildWatch {
    std::vector<pid_t> live;                          // (guarded by mu once start() has been called)
    std::mutex mu; std::thread th; int worst = 0; bool started = false;
    static constexpr int GRACE_MS = 5000;
    void kill_all(int sig = SIGTERM) { std::lock_guard<std::mutex> g(mu); for (pid_t p : live) kill(p, sig); }
    void start()
    {
        started = true;
        th = std::thread([this] {
            bool killing = false; auto deadline = std::chrono::steady_clock::now();
            for (;;) {
                { std::lock_guard<std::mutex> g(mu); if (live.empty()) return; }
                int st = 0;
                const pid_t pid = waitpid(-1, &st, killing ? WNOHANG : 0);
                if (pid < 0 && errno != EINTR) return;                        // no children left
                if (pid > 0) {
                    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
                    bool mine = false;
                    { std::lock_guard<std::mutex> g(mu); auto it = std::find(live.begin(), live.end(), pid); if (it != live.end()) { live.erase(it); mine = true; } }
                    if (!mine) continue;
                    if (code && !killing) {
                        worst = code;
                        fprintf(stderr, "DF: a rank process ended with status %d; stopping the others\n", code);
                        kill_all(SIGTERM);
                        killing = true; deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(GRACE_MS);
                    }
                    continue;
                }
                if (killing) {
                    if (std::chrono::steady_clock::now() > deadline) { kill_all(SIGKILL); deadline += std::chrono::hours(1); }
                    std::this_thread::sleep_for(std::chrono::milliseconds(20));
                }
            }
        });
    }
    int join() { if (started && th.joinable()) th.join(); return worst; }
    ~ChildWatch() { if (started && th.joinable()) { kill_all(SIGKILL); th.join(); } }
};

} // namespace

int main(int argc, char** argv)
{
    { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); printf("DF_MAIN_EPOCH %.3f\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec); }
    const double t_start = now_s();
    // the library's host transfers as this program was measured, unless the caller says otherwise: four lanes
    // per transfer (three writers run side by side in the last phase), eight for the uploads (which run alone)
    (void)setenv("DFK_HOST_THREADS", "4", 0); (void)setenv("DFK_UPLOAD_THREADS", "8", 0);
    std::map<std::string, std::string> a = {
        {"K", "48"}, {"MIN_FREQ", "3"}, {"MIN_BC", "2"}, {"MIN_QUAL", "7"}, {"ROOT", "/mnt/assembly"}, {"INSTANCE", "1"},
        {"OUT_DIR", ""}, {"LR", ""}, {"LR_SELECT_FRAC", "1.0"}, {"EXIT_LOAD", "False"}, {"DEVICE", "0"}, {"MAX_MEM_GB", "0"},
        {"HBM_GB", "0"}, {"NUM_THREADS", "-1"}, {"MINIMIZER", "0"}, {"KVEC", "Auto"}, {"KVEC_SORTED", "False"}, {"GRAPH", "True"}, {"PATHS", "True"}, {"LINK_READS", "False"}, {"NUM_GPUS", "1"}, {"PATHS_RESERVE", "20"}};
    std::string command = "DF";
    for (int i = 1; i < argc; ++i) {
        std::string s = argv[i]; command += " " + s;
        size_t eq = s.find('=');
        if (eq == std::string::npos) give_up("DF: arguments are KEY=VALUE; got '" + s + "'");
        a[s.substr(0, eq)] = s.substr(eq + 1);        // other reference arguments (PIPELINE, ALIGN, ...) are accepted and unused
    }
    const unsigned K = (unsigned)atoi(a["K"].c_str());
    if (K != 40 && K != 48 && K != 60) give_up("K must be 40, 48 or 60");                        // DF.cc:209
    if (a["LR"].empty()) give_up("I'm not sure you really want to do this, since it may\ndelete your starting files.  So I'm going to quit.");
    std::vector<double> select_frac;
    for (const std::string& f : parse_set(a["LR_SELECT_FRAC"])) select_frac.push_back(atof(f.c_str()));
    {   // NUM_THREADS: SetThreads (system/System.cc:1031-1046): <= 0 means all the machine has
        const int nt = atoi(a["NUM_THREADS"].c_str());
        g_threads = nt > 0 ? (unsigned)nt : std::max(1u, std::thread::hardware_concurrency());
        g_threads = std::min(g_threads, 64u);
    }

    std::string work_dir = a["ROOT"] + "/GapToy/" + a["INSTANCE"];                                  // DF.cc:221-222
    if (!a["OUT_DIR"].empty()) work_dir = a["OUT_DIR"];
    mkpath(work_dir); mkpath(work_dir + "/data"); mkpath(work_dir + "/stats"); mkpath(work_dir + "/logs");
    { FILE* f = fopen((work_dir + "/the_command").c_str(), "a"); if (f) { char h[256] = "host"; gethostname(h, sizeof h); fprintf(f, "\n%s: %s\n", h, command.c_str()); fclose(f); } }

    std::vector<std::string> heads;
    for (std::string lr : parse_set(a["LR"])) {
        size_t p = lr.find(".fastb");
        std::string head = p == std::string::npos ? lr : lr.substr(0, p);
        if (!is_file(head + ".fastb") || !is_file(head + ".qualp") || !is_file(head + ".bci"))
            give_up("Can't file your LR input files " + head + ".*.");                             // DF.cc:251-258
        heads.push_back(head);
    }
    // ---- NUM_GPUS > 1: one process per GPU, spawned HERE, before this process has made a single HIP call (a process that
    //      has initialised the GPU must not be replaced or forked).  The children (DF_RANK set) run the sharded createDict
    //      and write the spectrum and kmers.kvec; this process does the ingest and its output files meanwhile, then waits.
    //      DF_TRANSPORT=loopback: all ranks as threads of this process on ONE GPU, records moved by device copies -- the way
    //      the multi-rank driver is exercised where there is one GPU (RCCL refuses two ranks on one device).
    const int num_gpus = std::max(1, atoi(a["NUM_GPUS"].c_str()));
    const bool loopback = getenv("DF_TRANSPORT") && std::string(getenv("DF_TRANSPORT")) == "loopback";
    if ((num_gpus & (num_gpus - 1)) || num_gpus > 64) give_up("NUM_GPUS must be a power of two (the owner of a minimizer bucket is its low bits)");
    if (getenv("DF_RANK")) return rank_main(a, K, work_dir, heads[0], atoi(getenv("DF_RANK")), atoi(getenv("DF_WORLD")), nullptr);
    // EXIT_LOAD (DF.cc:483) stops after the ingest: no rank is spawned for it
    const bool sharded = (num_gpus > 1 || getenv("DF_FORCE_SHARDED")) && !truthy(a["EXIT_LOAD"]);
    auto mark = [&](const char* what) { if (getenv("DFK_TRACE")) fprintf(stderr, "[DF] %.3f s: %s\n", now_s() - t_start, what); };
    mark("arguments read");
    if (!sharded && !truthy(a["EXIT_LOAD"])) bind_to_gpu_node(atoi(a["DEVICE"].c_str()));      // (a parent of ranks must not touch the GPU before it forks)
    mark("bound to the GPU's node");
    ChildWatch watch;
    std::vector<pid_t>& children = watch.live;
    if (sharded) {
        // the ranks read the one input by pair range: LoadData's order must be the input's (one LR input, every pair kept)
        if (heads.size() != 1) give_up("NUM_GPUS > 1 takes one LR input");
        for (double f : select_frac) if (f < 1.0) give_up("NUM_GPUS > 1 needs LR_SELECT_FRAC = 1 (the ranks count the input as it is)");
        if (!loopback) {
            unlink((work_dir + "/.dfk_rccl_id").c_str());
            const pid_t parent_pid = getpid();
            for (int r = 0; r < num_gpus; ++r) {
                const pid_t pid = fork();
                if (pid < 0) { watch.kill_all(); give_up("cannot fork"); }
                if (pid == 0) {
                    // A rank must not outlive this process: ChildWatch covers a rank that dies, this covers the parent being
                    // killed outright (a harness timeout, the OOM killer) while the ranks sit in a collective that has no
                    // timeout and hold their GPUs.  Set before the exec (it survives it) and before any GPU call.
                    prctl(PR_SET_PDEATHSIG, SIGKILL);
                    if (getppid() != parent_pid) _exit(1);                   // (the parent went away between the fork and the prctl)
                    setenv("DF_RANK", std::to_string(r).c_str(), 1); setenv("DF_WORLD", std::to_string(num_gpus).c_str(), 1);
                    execv("/proc/self/exe", argv);
                    _exit(127);
                }
                children.push_back(pid);
            }
            watch.start();
        }
    }
    auto wait_children = [&]() -> int { return watch.join(); };
    Timing T;
    try {
        // ---- LoadData (10X/DfTools.cc:69-170): unbarcoded pairs of every input first, then barcoded pairs
        //      barcode by barcode; bci rebuilt; pairs stay together (even = R1, odd = R2).
        printf("%s: reading in linked read data\n", date().c_str());
        double t0 = now_s();
        struct In { FeudalMap fb, qp; std::vector<int64_t> bci; };
        std::vector<In> ins(heads.size());
        std::vector<std::thread> background;              // output files written while the GPU counts; checks nothing waits for
        std::exception_ptr bg_err; std::atomic<bool> bg_failed{false};
        struct Joiner { std::vector<std::thread>& v; ~Joiner() { for (auto& t : v) if (t.joinable()) t.join(); } } joiner{background};
        auto in_background = [&](std::function<void()> fn) {
            background.emplace_back([&, fn] { try { fn(); } catch (...) { if (!bg_failed.exchange(true)) bg_err = std::current_exception(); } });
        };
        auto join_background = [&] {
            for (auto& x : background) x.join();
            background.clear();
            if (bg_failed) std::rethrow_exception(bg_err);
        };
        for (size_t i = 0; i < heads.size(); ++i) {
            In& x = ins[i];
            x.fb.open(heads[i] + ".fastb", false);
            x.qp.open(heads[i] + ".qualp", false);
            x.bci = feudal::read_bci(heads[i] + ".bci");
            if (x.qp.n != x.fb.n) throw std::runtime_error(heads[i] + ": .fastb and .qualp disagree on the number of reads");
            if (x.fb.m.n - x.fb.fixedOff < 4 * x.fb.n) throw std::runtime_error(heads[i] + ".fastb: fixed data too short");
            if (x.bci.size() < 2 || x.bci[0] != 0) throw std::runtime_error("barcode 0 is unbarcoded data and must start at 0");
            if (x.bci[1] % 2 || (uint64_t)x.bci[1] > x.fb.n || (uint64_t)x.bci.back() > x.fb.n)
                throw std::runtime_error(heads[i] + ": .bci does not describe these reads");
            for (size_t b = 0; b + 1 < x.bci.size(); ++b)
                if (x.bci[b] > x.bci[b + 1]) throw std::runtime_error(heads[i] + ": .bci is not ascending");
            // .fastb's offset table in ONE pass: ascending, inside the data, and every read ceil(len/4) bytes -- what lets the count
            // derive the table on the device from the lengths.  .qualp's table is checked beside what follows: the device checks
            // it again before it reads anything through it (k_trim), the host's readers of it (the general path) wait for this one.
            const uint8_t* lens = x.fb.fixed();
            struct DenseSize { const uint8_t* lens; uint64_t operator()(uint64_t r) const { return ((uint64_t)ld32(lens + 4 * r) + 3) / 4; } };
            x.fb.check_offsets(DenseSize{lens}, ".fastb: read length disagrees with its byte count");
            FeudalMap* qp = &x.qp;
            in_background([qp] { qp->check_offsets(); });
        }
        if (select_frac.size() == 1 && heads.size() > 1) select_frac.assign(heads.size(), select_frac[0]);
        if (select_frac.size() != heads.size()) throw std::runtime_error("LR_SELECT_FRAC needs one value per LR input");   // DfTools.cc:96
        T.read = now_s() - t0;
        mark("inputs mapped and checked");

        // The arrays createDict sees.  Fast path (one input, every pair kept, every barcode an even number of reads:
        // what ParseBarcodedFastqs writes and runall.sh:127 passes): LoadData's order IS the input's order, so the
        // mapped files are the arrays -- var data addressed through the files' own absolute offset tables.
        bool fast = heads.size() == 1 && select_frac[0] >= 1.0;
        if (fast) for (size_t b = 0; b + 1 < ins[0].bci.size(); ++b) if ((ins[0].bci[b + 1] - ins[0].bci[b]) % 2) { fast = false; break; }
        if (fast && (uint64_t)ins[0].bci.back() != ins[0].fb.n) fast = false;
        if (!fast) join_background();                     // (the general path reads .qualp through its table on the host: checked first)
        feudal::Reads R;                                  // general path only
        std::vector<int64_t> bci;
        std::vector<DataSet> datasets;
        uint64_t n_reads = 0;
        const uint8_t *h_packed, *h_boff, *h_len, *h_pq, *h_qoff;   // (possibly unaligned: inside mapped files)
        const std::string rh = work_dir + "/data/frag_reads_orig";
        RefRandom rng;
        t0 = now_s();
        if (fast) {
            In& x = ins[0];
            n_reads = x.fb.n; bci = std::move(x.bci);                          // (80 MB at configs[1]: not copied)
            DataSet d{}; d.dt = 2; d.start = 0; datasets.push_back(d);
            d.dt = 3; d.start = bci[1]; datasets.push_back(d);
            h_packed = x.fb.m.p; h_boff = x.fb.off_table(); h_len = x.fb.fixed();
            h_pq = x.qp.m.p; h_qoff = x.qp.off_table();
            const bool may_link = truthy(a["LINK_READS"]);
            if (!(may_link && link_feudal(x.fb, rh + ".fastb", 4, 16, 1))) in_background([&] { copy_feudal(x.fb, rh + ".fastb", 4, 16, 1); });      // (two files, two writers)
            if (!(may_link && link_feudal(x.qp, rh + ".qualp", 0, 8, 1))) in_background([&] { copy_feudal(x.qp, rh + ".qualp", 0, 8, 1); });
            in_background([&bci, rh] { feudal::BinWriter w(rh + ".bci"); w.vec(bci); });     // (the count thread, which reads bci too, is not kept waiting for it)
        } else {
            R.base_off.push_back(0); R.pq_off.push_back(0);
            bci.push_back(0);
            // every pair asks the reference's random stream whether it stays (DfTools.cc:115-117): always at
            // LR_SELECT_FRAC = 1, but the number is drawn all the same, and WriteSubSample continues the stream
            auto append = [&](const In& x, double frac, int64_t lo, int64_t hi) {
                for (int64_t r = lo; r + 1 < hi + (hi - lo) % 2; r += 2) {
                    if (!((1. * rng.next() / 2147483647.0) <= frac)) continue;
                    for (int64_t k = r; k < std::min<int64_t>(r + 2, hi); ++k) {
                        const uint64_t b0 = x.fb.off(k), b1 = x.fb.off(k + 1), q0 = x.qp.off(k), q1 = x.qp.off(k + 1);
                        R.packed.insert(R.packed.end(), x.fb.m.p + b0, x.fb.m.p + b1);
                        R.pq.insert(R.pq.end(), x.qp.m.p + q0, x.qp.m.p + q1);
                        R.base_off.push_back(R.base_off.back() + (b1 - b0));
                        R.pq_off.push_back(R.pq_off.back() + (q1 - q0));
                        R.read_len.push_back(ld32(x.fb.fixed() + 4 * k));
                    }
                }
            };
            for (size_t i = 0; i < ins.size(); ++i) {                                                   // PASS_UNBARCODED
                DataSet d{}; d.dt = 2; d.start = (int64_t)R.size(); datasets.push_back(d);
                append(ins[i], select_frac[i], 0, ins[i].bci[1]);
            }
            for (size_t i = 0; i < ins.size(); ++i) {                                                   // PASS_BARCODED
                const In& x = ins[i];
                DataSet d{}; d.dt = 3; d.start = (int64_t)R.size(); datasets.push_back(d);
                for (size_t b = 1; b + 1 < x.bci.size(); ++b) { bci.push_back((int64_t)R.size()); append(x, select_frac[i], x.bci[b], x.bci[b + 1]); }
            }
            bci.push_back((int64_t)R.size());
            ins.clear();
            n_reads = R.size();
            h_packed = R.packed.data(); h_boff = (const uint8_t*)R.base_off.data(); h_len = (const uint8_t*)R.read_len.data();
            h_pq = R.pq.data(); h_qoff = (const uint8_t*)R.pq_off.data();
            feudal::write_fastb(rh + ".fastb", R.packed.data(), R.base_off, R.read_len);
            feudal::write_qualp(rh + ".qualp", R.pq.data(), R.pq_off);
            { feudal::BinWriter w(rh + ".bci"); w.vec(bci); }
        }
        auto boff = [&](uint64_t r) { return ld64(h_boff + 8 * r); };
        auto qoff = [&](uint64_t r) { return ld64(h_qoff + 8 * r); };
        auto rlen = [&](uint64_t r) { return ld32(h_len + 4 * r); };
        auto subsample = [&, n_reads] {
            // WriteSubSample(bases, quals, 500, ".../frag_reads_orig.1000") (DfTools.cc:32-67,169), on the same stream.
            // A pair is kept when its draw says so, or when only as many are left as are wanted.
            const uint64_t n = n_reads;
            uint64_t want = std::min<uint64_t>(n / 2, 500);
            const double frac = n / 2 ? (double)want / (double)(n / 2) : 0.0;
            std::vector<uint8_t> sp, sq; std::vector<uint64_t> so{0}, sqo{0}; std::vector<uint32_t> sl;
            for (uint64_t i = 0; i + 1 < n && want; i += 2) {
                const bool take = (1. * rng.next() / 2147483647.0) <= frac;
                if (!(take || want * 2 >= n - i)) continue;
                for (uint64_t k = i; k < i + 2; ++k) {
                    sp.insert(sp.end(), h_packed + boff(k), h_packed + boff(k + 1)); so.push_back(sp.size());
                    sq.insert(sq.end(), h_pq + qoff(k), h_pq + qoff(k + 1)); sqo.push_back(sq.size());
                    sl.push_back(rlen(k));
                }
                --want;
            }
            feudal::write_fastb(rh + ".1000.fastb", sp.data(), so, sl);
            feudal::write_qualp(rh + ".1000.qualp", sq.data(), sqo);
        };
        // fast path: the reference's random stream is advanced by one draw per pair in LoadData (all kept), then
        // WriteSubSample draws once more per pair -- 2 x 10^9 sequential draws at human scale, off the critical path
        if (fast) in_background([&, subsample] { for (uint64_t i = 0; i < n_reads / 2; ++i) (void)rng.next(); subsample(); });
        else subsample();
        printf("%s: loaded %llu reads\n", date().c_str(), (unsigned long long)n_reads);
        for (const DataSet& d : datasets) printf("\t%s starts at %ld\n", d.dt == 2 ? "UNBAR_10X" : "BAR_10X", (long)d.start);

        // ---- barcode expansion (DF.cc:447-452) and createDict on the GPU.  With the inputs mapped in place (fast path) the upload
        //      and the count start NOW, in a thread of their own, while this one computes the side files below; otherwise
        //      they run where the reference has them.
        std::unique_ptr<int32_t[]> bc;                                          // (not a vector: 7 GB need not be zeroed by one thread first)
        dfk_config cfg{};
        cfg.abi_version = DFK_ABI_VERSION; cfg.K = K; cfg.min_qual = (uint32_t)atoi(a["MIN_QUAL"].c_str());
        cfg.min_freq = (uint32_t)atoi(a["MIN_FREQ"].c_str()); cfg.min_bc = (uint32_t)atoi(a["MIN_BC"].c_str());
        cfg.device = atoi(a["DEVICE"].c_str()); cfg.ign_bc_below = 0;           // bc_start = 0 for LR-only input (DF.cc:344-349)
        cfg.minimizer_len = (uint32_t)atoi(a["MINIMIZER"].c_str());
        cfg.hbm_budget_bytes = (uint64_t)(atof(a["HBM_GB"].c_str()) * 1073741824.0);   // 0 = 90 % of the free HBM; MAX_MEM_GB is host memory
        const bool want_paths = truthy(a["GRAPH"]) && truthy(a["PATHS"]);
        // kmers.kvec is a transient file of the reference (written at BuildReadQGraph48.cc:287, read back at :294-301, removed
        // at :303): with the graph built here nothing downstream reads it, so it is written only on request -- or when
        // GRAPH=False, where it is the one way the dictionary leaves this process
        const bool want_kvec = a["KVEC"] == "Auto" ? (!truthy(a["GRAPH"]) || truthy(a["KVEC_SORTED"])) : truthy(a["KVEC"]);
        if (want_paths) cfg.flags |= DFK_F_KEEP_INPUTS;                          // the reads stay on the device for pathReads
        // a.<K>/a.paths and a.paths.inv are the stage's largest outputs (20-odd and 8 bytes a read: 37 and 14 GB at configs[1]) and
        // their sizes are known only once the reads are pathed.  Pages for them are made NOW (PATHS_RESERVE bytes a read for
        // a.paths -- 16 fixed and one path entry -- and 8 for every entry that leaves for a.paths.inv; 0 = don't), by threads nobody
        // waits for until the files are written: a pwrite into pages that exist does not allocate them under the file's lock.
        // The library sets the sizes; what was reserved beyond them is freed again (paid for twice, so the guess is a low one).
        std::thread reserve_thread, reserve_inv_thread;
        struct JoinReserve { std::thread& t; ~JoinReserve() { if (t.joinable()) t.join(); } } reserve_joiner{reserve_thread}, reserve_inv_joiner{reserve_inv_thread};
        const double reserve_per_read = atof(a["PATHS_RESERVE"].c_str());
        if (fast && !sharded && want_paths && !truthy(a["EXIT_LOAD"]) && reserve_per_read > 0 && n_reads) {
            const std::string dir = work_dir + "/a." + std::to_string(K);
            mkpath(dir);
            const uint64_t bytes = 24 + (uint64_t)((double)n_reads * reserve_per_read);
            auto reserve = [](const std::string& path, uint64_t n) {
                const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
                if (fd < 0) return;                                             // (the writer will say what is wrong with the path)
                if (fallocate(fd, 0, 0, (off_t)n) != 0) (void)!ftruncate(fd, 0);            // no room, or a file system without it: the file grows as it is written
                close(fd);
            };
            const uint64_t inv_bytes = 24 + (uint64_t)((double)n_reads * std::max(0.0, reserve_per_read - 16.0) * 2.0);
            reserve_thread = std::thread([=] { reserve(dir + "/a.paths", bytes); });
            reserve_inv_thread = std::thread([=] { if (inv_bytes > 24) reserve(dir + "/a.paths.inv", inv_bytes); });
        }
        dfk_ctx* ctx = nullptr;
        int count_rc = 0; bool create_failed = false; std::string count_err; double t_count = 0;
        // frag_reads_orig.qhist: counted on the device from the reads the count keeps there (the pathing wants them), behind the
        // count, once this thread has been through the read lengths; counted here when the device keeps nothing
        const bool device_hist = fast && !sharded && !truthy(a["EXIT_LOAD"]) && want_paths && !getenv("DF_HOST_QHIST");
        std::atomic<int> max_len_known{-1};
        std::vector<int64_t> qh_dev; std::string qh_err; double t_qhist = 0;
        // The barcode expansion of DF.cc:447-452 (7 GB of int32 at configs[1]) and the .fastb offset table (14 GB: a running sum of
        // the read lengths, as the validation above has just established) are made on the device from the 80 MB index and the
        // lengths (dfk_count_bci, base_off = NULL): neither is written by the host or crosses PCIe.  DF_HOST_BC=1: as before.
        const bool host_bc = getenv("DF_HOST_BC") != nullptr;
        auto count_job = [&] {
            const double tj0 = now_s();
            if (host_bc) {
                bc.reset(new int32_t[std::max<uint64_t>(1, n_reads)]);
                parallel_ranges(n_reads, [&](unsigned, uint64_t lo, uint64_t hi) { memset(bc.get() + lo, 0, 4 * (hi - lo)); });      // (reads no barcode's range holds: 0)
                parallel_ranges(bci.size() - 1, [&](unsigned, uint64_t lo, uint64_t hi) {
                    for (uint64_t b = lo; b < hi; ++b) for (int64_t r = bci[b]; r < bci[b + 1]; ++r) bc[r] = (int32_t)b;
                }, 1024);
            }
            const double tj1 = now_s();
            if (dfk_create(&cfg, &ctx)) { create_failed = true; count_err = dfk_last_error(); return; }
            if (getenv("DFK_TRACE")) fprintf(stderr, "[DF] barcodes expanded in %.3f s, dfk_create %.3f s, %.3f s after the process started\n", tj1 - tj0, now_s() - tj1, now_s() - t_start);
            if (fast)                                                           // the arrays are maps of these files: what has been uploaded leaves the page table (see below)
                for (const Mapped* m : {&ins[0].fb.m, &ins[0].qp.m})
                    if (m->p && dfk_hint_file_range(ctx, m->p, m->n, -1, 0)) { create_failed = true; count_err = dfk_last_error(); return; }
            printf("%s: building dictionary on the GPU\n", date().c_str());
            const double tc = now_s();
            if (host_bc) count_rc = dfk_count(ctx, h_packed, (const uint64_t*)h_boff, (const uint32_t*)h_len, h_pq, (const uint64_t*)h_qoff, bc.get(), n_reads);
            else if (fast) count_rc = dfk_count_bci(ctx, h_packed + (n_reads ? ld64(h_boff) : 0), nullptr, (const uint32_t*)h_len, h_pq, (const uint64_t*)h_qoff, bci.data(), bci.size(), n_reads);   // (dense: validated above)
            else count_rc = dfk_count_bci(ctx, h_packed, (const uint64_t*)h_boff, (const uint32_t*)h_len, h_pq, (const uint64_t*)h_qoff, bci.data(), bci.size(), n_reads);
            if (count_rc) count_err = dfk_last_error();
            t_count = now_s() - tc;
            if (!count_rc && device_hist) {
                while (max_len_known.load() < 0) std::this_thread::sleep_for(std::chrono::milliseconds(2));
                const int ml = max_len_known.load();
                qh_dev.assign((size_t)2 * ml * 256, 0);
                const double tq = now_s();
                if (ml && dfk_qual_hist(ctx, (uint32_t)ml, qh_dev.data())) qh_err = dfk_last_error();
                t_qhist = now_s() - tq;
            }
        };
        std::thread count_thread;
        struct JoinAtExit { std::thread& t; ~JoinAtExit() { if (t.joinable()) t.join(); } } count_joiner{count_thread};   // (an exception below must not leave it running)
        struct LenAtExit { std::atomic<int>& v; ~LenAtExit() { if (v.load() < 0) v = 0; } } len_guard{max_len_known};   // (the count thread waits for it)
        mark("count thread starts");
        if (fast && !sharded && !truthy(a["EXIT_LOAD"])) count_thread = std::thread(count_job);

        // ---- lens, quality histogram, datasets (DF.cc:50-68, DfTools.cc:172-238)
        std::vector<int16_t> lens(n_reads);
        std::vector<int> tmax(g_threads + 1, 0);
        parallel_ranges(n_reads, [&](unsigned t, uint64_t lo, uint64_t hi) {
            int m = 0;
            for (uint64_t i = lo; i < hi; ++i) { lens[i] = (int16_t)rlen(i); m = std::max<int>(m, lens[i]); }
            tmax[t] = m;
        }, 1 << 16, true);
        const int max_len = *std::max_element(tmax.begin(), tmax.end());
        max_len_known = max_len;
        printf("%s: computing quality histogram\n", date().c_str());
        std::vector<int64_t> qh((size_t)2 * max_len * 256, 0);                  // [parity][pos][q]
        if (!device_hist) {
            std::vector<std::unique_ptr<QualHist>> part(g_threads + 1);
            parallel_ranges(n_reads, [&](unsigned t, uint64_t lo, uint64_t hi) {
                part[t].reset(new QualHist(max_len));
                for (uint64_t r = lo; r < hi; ++r) part[t]->add_read(r, h_pq + qoff(r), h_pq + qoff(r + 1));
            }, 1 << 16, true);
            for (const auto& p : part) if (p) p->merge_into(qh);
        }
        auto write_qhist = [&](const std::vector<int64_t>& h) {
            int max_q = -1;
            for (size_t i = 0; i < h.size(); ++i) if (h[i]) max_q = std::max(max_q, (int)(i & 255));
            feudal::BinWriter w(rh + ".qhist");                                 // vec<vec<vec<int64_t>>> [2][max_len][max_q+1]
            w.pod<uint64_t>(2);
            for (int par = 0; par < 2; ++par) {
                w.pod<uint64_t>((uint64_t)max_len);
                for (int pos = 0; pos < max_len; ++pos) { w.pod<uint64_t>((uint64_t)(max_q + 1)); w.raw(&h[((size_t)par * max_len + pos) * 256], 8 * (size_t)(max_q + 1)); }
            }
        };
        { feudal::BinWriter w(rh + ".lens"); w.vec(lens); }
        if (!device_hist) write_qhist(qh);
        { std::vector<int16_t>().swap(lens); }
        { feudal::BinWriter w(rh + ".dti"); w.vec(datasets); }
        { feudal::BinWriter w(work_dir + "/subsam.names"); w.pod<uint64_t>(1); w.str("C"); }
        { feudal::BinWriter w(work_dir + "/subsam.starts"); w.vec(std::vector<int64_t>{0}); }
        T.ingest_out = now_s() - t0;
        if (truthy(a["EXIT_LOAD"])) { join_background(); return 0; }             // DF.cc:483
        if (sharded) {
            // the ranks are counting (or, loopback, start now); this process has done the ingest
            int rc = 0;
            if (!fast) throw std::runtime_error("NUM_GPUS > 1 needs one LR input with LR_SELECT_FRAC = 1 and whole pairs per barcode");
            if (loopback) {
                dfkx::LoopbackHub hub(num_gpus, [](void* d, const void* s_, uint64_t nbytes) { if (hipMemcpy(d, s_, nbytes, hipMemcpyDeviceToDevice) != hipSuccess) throw std::runtime_error("device copy failed"); });
                std::vector<int> rcs(num_gpus, 0);
                std::vector<std::thread> th;
                for (int r = 0; r < num_gpus; ++r) th.emplace_back([&, r] { rcs[r] = rank_main(a, K, work_dir, heads[0], r, num_gpus, &hub); });
                for (auto& x : th) x.join();
                for (int r : rcs) rc = std::max(rc, r);
            } else rc = wait_children();
            join_background();
            T.total = now_s() - t_start;
            printf("%s: ingest + sharded count stage %.2f s wall on %d GPUs\n", date().c_str(), T.total, num_gpus);
            return rc;
        }

        // ---- createDict on the GPU: started above (fast path) or here
        if (!count_thread.joinable()) count_job();
        else count_thread.join();
        if (create_failed) { fprintf(stderr, "DF: %s\n", count_err.c_str()); join_background(); return 1; }
        const int rc = count_rc;
        if (!rc && device_hist) {
            if (!qh_err.empty()) throw std::runtime_error(qh_err);
            write_qhist(qh_dev);
            std::vector<int64_t>().swap(qh_dev);
        }
        if (rc == DFK_E_NOGOOD) { printf("\nLooks like your input data have almost no good bases.\nGiving up.\n\n"); join_background(); return 1; }   // :227-230
        if (rc) { fprintf(stderr, "DF: %s\n", count_err.c_str()); join_background(); return rc == DFK_E_NOMEM ? 185 : 1; }        // Martian::exit code
        dfk_stats st{}; dfk_get_stats(ctx, &st);
        T.upload = 1e-3 * st.ms_upload; T.count = t_count - T.upload;
        bc.reset();
        if (fast) {
            // The mapped inputs are on the device (and stay there for the pathing), and the transfer lanes have dropped what they
            // copied from the page table (the hint above): unmapping them is cheap now, where it would have been three seconds
            // of one thread's page-table work -- at exit, if not before.  Behind the tasks that still read the maps.
            auto earlier = std::make_shared<std::vector<std::thread>>(std::move(background));
            background.clear();
            background.emplace_back([&ins, earlier] { for (auto& x : *earlier) x.join(); for (In& x : ins) { x.fb.m.unmap(); x.qp.m.unmap(); } });
        }
        mark("count joined");
        t0 = now_s();
        uint64_t need = 0; dfk_spectrum_json(ctx, nullptr, 0, &need);
        std::string js(need, '\0'); dfk_spectrum_json(ctx, &js[0], need, &need);
        { FILE* f = fopen((work_dir + "/stats/histogram_kmer_count.json").c_str(), "wb"); if (!f) throw std::runtime_error("cannot write spectrum"); fwrite(js.data(), 1, js.size(), f); fclose(f); }
        uint64_t nk = 0; dfk_solid_count(ctx, &nk);
        if (want_kvec) {
            printf("%s: writing kmers.kvec\n", date().c_str());
            if (dfk_write_kvec(ctx, (work_dir + "/kmers.kvec").c_str(), truthy(a["KVEC_SORTED"]) ? DFK_KVEC_SORTED : 0)) throw std::runtime_error(dfk_last_error());
        }
        T.fetch_write = now_s() - t0;
        // ---- buildEdges + buildHBVFromEdges + the graph files of WriteAssemblyFiles (BuildReadQGraph48.cc:1636,1664;
        //      10X/WriteFiles.cc:69-101): a.<K>/{a.k,a.hbv,a.hbx,a.to_left,a.to_right,a.inv,a.fastb,a.kmers}
        double t_graph = 0, t_g_dev = 0, t_g_host = 0, t_g_write = 0, t_paths = 0, t_p_dev = 0, t_p_write = 0, t_index = 0, t_dups = 0;
        uint64_t g_ce = 0, g_v = 0, g_e = 0, p_placed = 0, p_edges = 0, n_dup = 0;
        if (truthy(a["GRAPH"])) {
            t0 = now_s();
            mark("graph starts");
            printf("%s: finding edge sequences.\n", date().c_str());
            if (dfk_graph_build(ctx)) throw std::runtime_error(dfk_last_error());
            { dfk_stats gs{}; dfk_get_stats(ctx, &gs); t_g_dev = 1e-6 * (double)gs.reserved[1]; t_g_host = 1e-6 * (double)gs.reserved[2]; }
            const std::string dir = work_dir + "/a." + std::to_string(K);
            mkpath(dir);
            printf("%s: writing files\n", date().c_str());
            // the graph's files are written (host work: numbering is done, the edges are packed) while the device paths the reads;
            // a.paths is written while the device inverts the paths and marks duplicates.  Both writers only read the context.
            std::string bg_fail; double tw_graph = 0, tw_paths = 0;
            std::thread graph_writer([&] { const double t1 = now_s(); if (dfk_graph_write(ctx, dir.c_str())) bg_fail = dfk_last_error(); tw_graph = now_s() - t1; });
            struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_graph{graph_writer};
            dfk_graph_stats(ctx, &g_ce, &g_v, &g_e);
            if (!want_paths) { graph_writer.join(); if (!bg_fail.empty()) throw std::runtime_error(bg_fail); }
            t_graph = now_s() - t0;
            if (want_paths) {
                // pathReads (BuildReadQGraph48.cc:1664-1665) and a.<K>/a.paths (10X/WriteFiles.cc:78-82)
                t0 = now_s();
                mark("pathing starts");
                printf("%s: pathing reads\n", date().c_str());
                if (reserve_thread.joinable()) reserve_thread.join();
                if (dfk_paths_sink(ctx, (dir + "/a.paths").c_str())) throw std::runtime_error(dfk_last_error());      // (its data goes to the file batch by batch)
                if (dfk_paths_build(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, 0)) throw std::runtime_error(dfk_last_error());
                { dfk_stats ps{}; dfk_get_stats(ctx, &ps); t_p_dev = 1e-6 * (double)ps.reserved[3]; }
                mark("pathing done");
                printf("%s: writing paths\n", date().c_str());
                std::string bg_fail2;
                std::thread paths_writer([&] { const double t1 = now_s(); if (dfk_paths_write(ctx, (dir + "/a.paths").c_str())) bg_fail2 = dfk_last_error(); tw_paths = now_s() - t1; });
                Join join_paths{paths_writer};
                dfk_paths_stats(ctx, nullptr, &p_placed, &p_edges);
                // writePathsIndex and MarkDups, the two steps DF takes right after StageBuildGraph (10X/DF.cc:550,560)
                const double ti = now_s();
                printf("%s: inverting paths index\n", date().c_str());
                if (reserve_inv_thread.joinable()) reserve_inv_thread.join();
                if (getenv("DF_INDEX_THEN_DUPS")) {                             // (one after the other, each on its own clock)
                    if (dfk_paths_index_write(ctx, dir.c_str())) throw std::runtime_error(dfk_last_error());
                    t_index = now_s() - ti;
                    const double td = now_s();
                    if (dfk_dups_write(ctx, (dir + "/a.dup").c_str(), &n_dup)) throw std::runtime_error(dfk_last_error());
                    t_dups = now_s() - td;
                } else {                                                        // a.paths.inv's lists go to the file while the duplicates are marked
                    if (dfk_paths_index_dups_write(ctx, dir.c_str(), (dir + "/a.dup").c_str(), &n_dup)) throw std::runtime_error(dfk_last_error());
                    t_index = now_s() - ti;                                     // (both: mark_dups_s stays 0)
                }
                mark("index and duplicate marks done");
                paths_writer.join();
                mark("a.paths written");
                graph_writer.join();
                mark("graph files written");
                if (!bg_fail.empty()) throw std::runtime_error(bg_fail);
                if (!bg_fail2.empty()) throw std::runtime_error(bg_fail2);
                t_p_write = tw_paths;
                t_paths = now_s() - t0;
                printf("%.2f%% of pairs appear to be duplicates\n", n_reads ? 100.0 * (double)n_dup / (double)(n_reads / 2) : 0.0);
                print_digests(ctx);
            }
            t_g_write = tw_graph;
        }
        const double t_d0 = now_s();
        // (the context is not taken apart when the process is about to leave by _exit, below: handing 270 GB back to the driver
        // buffer by buffer takes 1.5 s, the operating system's own teardown of the process a fraction of that)
        const bool quick_exit = !sharded && !getenv("DF_SLOW_EXIT");
        if (!quick_exit) dfk_destroy(ctx);
        const double t_destroy = now_s() - t_d0;
        join_background();
        mark("background joined");
        const double t_joined = now_s() - t_d0 - t_destroy;
        T.total = now_s() - t_start;
        printf("%s: dictionary covers %llu kmers\n", date().c_str(), (unsigned long long)nk);
        printf("%s: %llu k-mer instances, GPU %.1f ms (count kernel %.1f ms), ingest+count stage %.2f s wall\n", date().c_str(),
               (unsigned long long)st.n_inst, st.ms_total, st.ms_count, T.total);
        // one machine-readable line (bench.py reads it): where the stage's wall time went
        printf("DF_TIMING {\"reads\": %llu, \"kmer_instances\": %llu, \"solid\": %llu, \"threads\": %u, \"open_validate_s\": %.3f, "
               "\"ingest_outputs_s\": %.3f, \"upload_s\": %.3f, \"count_s\": %.3f, \"spectrum_kvec_write_s\": %.3f, \"graph_s\": %.3f, \"graph_device_s\": %.3f, \"graph_host_s\": %.3f, \"graph_write_s\": %.3f, "
               "\"graph_edges\": %llu, \"graph_vertices\": %llu, \"paths_s\": %.3f, \"paths_device_s\": %.3f, \"paths_write_s\": %.3f, \"reads_placed\": %llu, \"path_edges\": %llu, \"paths_index_s\": %.3f, \"mark_dups_s\": %.3f, \"dup_pairs\": %llu, \"qual_hist_s\": %.3f, \"destroy_s\": %.3f, \"background_join_s\": %.3f, \"total_s\": %.3f, \"fast_path\": %s}\n",
               (unsigned long long)n_reads, (unsigned long long)st.n_inst, (unsigned long long)nk, g_threads, T.read, T.ingest_out,
               T.upload, T.count, T.fetch_write, t_graph, t_g_dev, t_g_host, t_g_write, (unsigned long long)g_e, (unsigned long long)g_v,
               t_paths, t_p_dev, t_p_write, (unsigned long long)p_placed, (unsigned long long)p_edges, t_index, t_dups, (unsigned long long)n_dup, t_qhist, t_destroy, t_joined, T.total, fast ? "true" : "false");
        { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); printf("DF_EXIT_EPOCH %.3f\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec); }
        // Everything is written and closed, the context destroyed, the ranks reaped: what is left is giving gigabytes of vectors
        // back to the allocator one by one and the HIP runtime's own teardown, none of which the operating system needs done
        // before it takes the process apart anyway.  (DF_SLOW_EXIT=1: the long way, for leak checkers.)
        if (quick_exit) { fflush(nullptr); _exit(0); }
    } catch (const std::exception& e) {
        fprintf(stderr, "DF: %s\n", e.what());
        watch.kill_all();
        (void)wait_children();
        return 1;
    }
    return 0;
}
