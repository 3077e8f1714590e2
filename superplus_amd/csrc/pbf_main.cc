// superplus_amd/csrc/pbf_main.cc -- `ParseBarcodedFastqs`: stLFR fastq.gz pairs -> barcode-sorted
// OUT_HEAD.{fastb,qualp,bci}, the files DF reads (runall.sh:125).  Host-only ingest, written from the behaviour of
// 10X/ParseBarcodedFastqs.cc:306-539 + mergeBarcodedReadFiles :222-304, not from its structure:
//
//   the reference reads both gz files once per barcode BUCKET (NUM_BUCKETS passes over the input), keeps each
//   barcode's pairs in a list it insertion-sorts (quadratic per barcode), writes one temporary file set per bucket
//   and merges them; here the files are read ONCE (one thread per file), pairs are grouped by barcode, each group is
//   stable-sorted, qualities are encoded on all threads and the three files are written directly.
//
// What must come out the same, byte for byte (checked against the reference's own binary, oracle/_ref/
// ParseBarcodedFastqs, built single-threaded -- a threaded reference run appends the buckets in the order its threads
// finish, i.e. in no reproducible barcode order):
//   * read names "@id#b1_b2_b3/1": barcode = b1*1537^2 + b2*1537 + b3 (10X/Barcode.cc:3-13); 0 = unbarcoded
//   * N -> A (:407-412); qualities = char - 33 (convertPhred :47-54)
//   * unbarcoded pairs first, in file order; then the buckets in order, inside a bucket the barcodes ascending
//     (std::set, :346), inside a barcode the pairs DEScending by (read 1, read 2) as base-code sequences, equal pairs
//     in file order (the insertion rule of :434-449)
//   * the buckets: the distinct barcodes in the iteration order of a std::unordered_set<int64_t> filled in file order,
//     cut into runs of floor(n / NUM_BUCKETS), the remainder joined to the last run (:318-336) -- the same standard
//     container gives the same order
//   * .qualp: PQVecEncoder's block choice (feudal/PQVec.cc:18-85, restated in pq_encode below)
//   * .bci: BINWRITE | u64 count | i64 offsets: 0, #unbarcoded reads, then the end of every barcode
#include "feudal_io.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <map>
#include <set>
#include <sstream>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <sys/stat.h>
#include <zlib.h>

namespace {

[[noreturn]] void die(const std::string& m) { fprintf(stderr, "ParseBarcodedFastqs: %s\n", m.c_str()); exit(1); }

std::vector<std::string> parse_set(std::string s)
{
    s.erase(std::remove(s.begin(), s.end(), '{'), s.end());
    s.erase(std::remove(s.begin(), s.end(), '}'), s.end());
    std::vector<std::string> out; std::stringstream ss(s); std::string tok;
    while (std::getline(ss, tok, ',')) if (!tok.empty()) out.push_back(tok);
    return out;
}

struct Fastq {                               // one file: per read its barcode, base codes and qualities
    std::vector<int64_t> bc;
    std::vector<uint8_t> bases, quals;       // concatenated
    std::vector<uint64_t> off{0};            // [n+1] into bases / quals
    std::string error;
    uint64_t counted = 0;                    // bases already added to g_held
};

// "...#b1_b2_b3/1\t..." -> b1*1537^2 + b2*1537 + b3  (10X/Barcode.cc:3-13: three integers, one separator character each)
bool barcode_of(const char* line, int64_t* out)
{
    const char* h = strrchr(line, '#');
    const char* p = h ? h + 1 : line;
    char* e;
    const long a = strtol(p, &e, 10); if (e == p || !*e) return false;
    p = e + 1;
    const long b = strtol(p, &e, 10); if (e == p || !*e) return false;
    p = e + 1;
    const long c = strtol(p, &e, 10); if (e == p) return false;
    *out = (int64_t)a * 1537 * 1537 + (int64_t)b * 1537 + c;
    return true;
}

// Memory: this program holds BOTH decompressed files (a byte per base and a byte per quality) and then the encoded reads
// twice while it writes -- about 3.5 bytes per base plus ~100 per read, where the reference bounds its own use by
// re-reading the inputs once per barcode bucket (10X/ParseBarcodedFastqs.cc:434-449).  g_mem_limit (MAX_MEM_GB, or the
// machine's memory) is checked while the files are read and before the encode: a set that does not fit ends with a
// message, not with the kernel's OOM killer.
std::atomic<uint64_t> g_held{0};
uint64_t g_mem_limit = 0;
bool over_budget(uint64_t bases, uint64_t reads) { return g_mem_limit && 7 * bases / 2 + 100 * reads > g_mem_limit; }
std::string budget_message(uint64_t bases, uint64_t reads)
{
    return "this input needs about " + std::to_string((7 * bases / 2 + 100 * reads) >> 30) + " GiB (3.5 bytes per base + 100 per read: the decompressed reads are held in "
           "memory), more than the " + std::to_string(g_mem_limit >> 30) + " GiB allowed (MAX_MEM_GB, or the machine's memory); run it on a machine with more memory or split the input";
}

void read_fastq(const std::string& path, Fastq* f)
{
    gzFile g = gzopen(path.c_str(), "rb");
    if (!g) { f->error = "cannot open " + path; return; }
    gzbuffer(g, 1 << 20);
    std::vector<char> line(1 << 16);
    auto get = [&]() -> bool { return gzgets(g, line.data(), (int)line.size()) != nullptr; };
    while (get()) {
        if (line[0] != '@') { f->error = "out of sync reading line: " + path + ": " + line.data(); break; }
        int64_t bc;
        if (!barcode_of(line.data(), &bc)) { f->error = "cannot parse the barcode of " + std::string(line.data()); break; }
        if (!get()) { f->error = "truncated record in " + path; break; }
        for (const char* p = line.data(); *p && *p != '\n' && *p != '\r'; ++p) {
            uint8_t v;
            switch (*p) { case 'A': case 'a': case 'N': case 'n': v = 0; break; case 'C': case 'c': v = 1; break;
                          case 'G': case 'g': v = 2; break; case 'T': case 't': v = 3; break;
                          default: f->error = std::string("unexpected base '") + *p + "' in " + path; v = 0; }
            f->bases.push_back(v);
        }
        if (!get() || !get()) { f->error = "truncated record in " + path; break; }       // '+' line, then the qualities
        size_t nq = 0;
        for (const char* p = line.data(); *p; ++p) if (*p != '\n' && *p != '\r') { f->quals.push_back((uint8_t)(*p - 33)); ++nq; }
        if (nq != f->bases.size() - f->off.back()) { f->error = "a read of " + path + " has " + std::to_string(nq) + " qualities for " + std::to_string(f->bases.size() - f->off.back()) + " bases"; break; }
        f->bc.push_back(bc);
        f->off.push_back(f->bases.size());
        if (!f->error.empty()) break;
        if ((f->bc.size() & 0xFFFF) == 0) {                                           // (both reader threads add to one total)
            const uint64_t mine = f->bases.size();
            const uint64_t all = g_held.fetch_add(mine - f->counted) + (mine - f->counted);
            f->counted = mine;
            if (over_budget(all, 0)) { f->error = "stopped reading " + path + ": " + budget_message(all, 0); break; }
        }
    }
    gzclose(g);
}

unsigned ceil_lg2(unsigned v) { unsigned b = 0; while ((1u << b) < v) ++b; return b; }
unsigned block_size(unsigned n, unsigned bits) { return (n * bits + 17 + 7) >> 3; }

// PQVecEncoder (feudal/PQVec.cc:18-127).  For every prefix the cheapest LAST block is chosen (1..255 values, cost = best
// cost of the prefix before it + the block's bytes, the shortest block winning ties); the block list of the longer
// prefix is the previous list cut back by the values the new block swallows, plus the new block.  (This is not a
// backtrace of the optimum -- the cut-back list need not be the best encoding of what remains -- so the procedure, not
// just its objective, is what has to be reproduced.)
struct PqBlock { uint8_t n, bits, minq; };
void pq_encode(const uint8_t* q, uint32_t len, std::vector<unsigned>& cost, std::vector<PqBlock>& blocks, std::vector<uint8_t>* out)
{
    cost.assign(1, 1); blocks.clear();
    for (uint32_t i = 0; i < len; ++i) {
        if (q[i] > 63) die("Your input reads are funny.  I found a quality score of " + std::to_string(q[i]) + ". The maximum value that I allow is 63.");
        unsigned mn = q[i], mx = q[i], bits = 0, n = 1;
        unsigned best_cost = cost[i] + block_size(1, 0);
        PqBlock best{1, 0, (uint8_t)mn};
        for (uint32_t j = i; j > 0 && n < 255;) {
            const unsigned v = q[--j];
            mx = std::max(mx, v); mn = std::min(mn, v);
            bits = ceil_lg2(mx + 1u - mn);
            const unsigned c = cost[j] + block_size(++n, bits);
            if (c < best_cost) { best_cost = c; best = PqBlock{(uint8_t)n, (uint8_t)bits, (uint8_t)mn}; }
        }
        cost.push_back(best_cost);
        unsigned remove = best.n - 1u;
        if (!remove) blocks.push_back(best);
        else {
            while (remove > blocks.back().n) { remove -= blocks.back().n; blocks.pop_back(); }
            if (remove == blocks.back().n) blocks.back() = best;
            else { blocks.back().n = (uint8_t)(blocks.back().n - remove); blocks.push_back(best); }
        }
    }
    const uint8_t* it = q;
    for (const PqBlock& b : blocks) {
        out->push_back(b.n);
        uint64_t acc = (uint64_t)b.bits | ((uint64_t)b.minq << 3);
        out->push_back((uint8_t)acc); acc >>= 8;
        if (!b.bits) { out->push_back((uint8_t)acc); it += b.n; continue; }
        unsigned off = 1;
        for (unsigned k = 0; k < b.n; ++k) {
            acc |= (uint64_t)(*it++ - b.minq) << off;
            if ((off += b.bits) >= 8) { out->push_back((uint8_t)acc); off -= 8; acc >>= 8; }
        }
        if (off) out->push_back((uint8_t)acc);
    }
    out->push_back(0);
}

} // namespace

int main(int argc, char** argv)
{
    std::map<std::string, std::string> a = {{"FASTQS", ""}, {"OUT_HEAD", ""}, {"NUM_BUCKETS", "256"}, {"READS_PER_BC", "0"},
                                            {"NUM_THREADS", "0"}, {"MAX_MEM_GB", "0"}, {"MERGE_HEADS", ""}};
    for (int i = 1; i < argc; ++i) {
        const std::string s = argv[i]; const size_t eq = s.find('=');
        if (eq == std::string::npos) die("arguments are KEY=VALUE; got '" + s + "'");
        a[s.substr(0, eq)] = s.substr(eq + 1);
    }
    if (a["OUT_HEAD"].empty()) die("OUT_HEAD is required");
    if (a["OUT_HEAD"].back() == '/') die("OUT_HEAD can not end with '/'");
    if (!a["MERGE_HEADS"].empty()) die("MERGE_HEADS is not supported: this program never writes per-bucket files to merge");
    const std::vector<std::string> fq = parse_set(a["FASTQS"]);
    if (fq.size() != 2) die("FASTQS must name two files: {R1.fq.gz,R2.fq.gz}");
    for (const std::string& f : fq) if (f.size() < 3 || f.compare(f.size() - 3, 3, ".gz")) die("read pair fastq input has to be in gz format");
    size_t n_buckets = (size_t)atoll(a["NUM_BUCKETS"].c_str());
    if (n_buckets == 0 || n_buckets > 256) die("NUM_BUCKETS must be in 1..256");
    const size_t reads_per_bc = (size_t)atoll(a["READS_PER_BC"].c_str());
    unsigned threads = (unsigned)atoi(a["NUM_THREADS"].c_str());
    if (!threads) threads = std::max(1u, std::thread::hardware_concurrency());
    threads = std::min(threads, 64u);

    {   // MAX_MEM_GB as the reference means it (a cap on host memory, system/System.cc:1073-1078); 0 = what the machine has
        const double gb = atof(a["MAX_MEM_GB"].c_str());
        const uint64_t phys = (uint64_t)sysconf(_SC_PHYS_PAGES) * (uint64_t)sysconf(_SC_PAGE_SIZE);
        g_mem_limit = gb > 0 ? std::min<uint64_t>(phys, (uint64_t)(gb * 1073741824.0)) : phys;
    }
    // ---- both files, one thread each
    Fastq f1, f2;
    { std::thread t(read_fastq, fq[1], &f2); read_fastq(fq[0], &f1); t.join(); }
    if (!f1.error.empty()) die(f1.error);
    if (!f2.error.empty()) die(f2.error);
    if (f1.bc.size() != f2.bc.size()) die("something not match with pair file: " + fq[0] + " or " + fq[1]);
    const size_t n_pairs = f1.bc.size();
    for (size_t i = 0; i < n_pairs; ++i) if (f1.bc[i] != f2.bc[i]) die("something not match with pair file: " + fq[0] + " or " + fq[1]);
    fprintf(stderr, "total reads: %zu\n", 2 * n_pairs);
    if (over_budget(f1.bases.size() + f2.bases.size(), 2 * n_pairs)) die(budget_message(f1.bases.size() + f2.bases.size(), 2 * n_pairs));

    // ---- buckets of barcodes (:311-336): the distinct barcodes in the container's iteration order, cut into equal runs
    std::unordered_set<int64_t> bc_set;
    for (size_t i = 0; i < n_pairs; ++i) bc_set.emplace(f1.bc[i]);
    fprintf(stderr, "total barcodes: %zu\n", bc_set.size());
    std::vector<std::vector<int64_t>> buckets;
    if (!bc_set.empty()) {
        if (n_buckets > bc_set.size()) n_buckets = bc_set.size();
        const size_t per = bc_set.size() / n_buckets;
        std::vector<int64_t> cur;
        for (int64_t b : bc_set) {
            if (cur.size() < per) cur.push_back(b);
            else { buckets.push_back(cur); cur.assign(1, b); }
        }
        if (!cur.empty()) {
            if (buckets.empty()) buckets.push_back(cur);
            else buckets.back().insert(buckets.back().end(), cur.begin(), cur.end());
        }
    }
    // ---- pairs grouped by barcode, in file order
    std::unordered_map<int64_t, std::vector<uint32_t>> group;
    for (size_t i = 0; i < n_pairs; ++i) group[f1.bc[i]].push_back((uint32_t)i);
    auto seq = [](const Fastq& f, uint32_t i) { return std::make_pair(f.bases.data() + f.off[i], f.bases.data() + f.off[i + 1]); };
    auto pair_greater = [&](uint32_t x, uint32_t y) {                  // (read 1, read 2) of x above those of y, as base-code sequences
        const auto x1 = seq(f1, x), y1 = seq(f1, y);
        if (std::lexicographical_compare(y1.first, y1.second, x1.first, x1.second)) return true;
        if (std::lexicographical_compare(x1.first, x1.second, y1.first, y1.second)) return false;
        const auto x2 = seq(f2, x), y2 = seq(f2, y);
        return std::lexicographical_compare(y2.first, y2.second, x2.first, x2.second);
    };
    // ---- output order of the pairs and the barcode index
    std::vector<uint32_t> order; order.reserve(n_pairs);
    std::vector<int64_t> bci{0};
    if (group.count(0)) order = group[0];
    bci.push_back((int64_t)(2 * order.size()));
    {
        std::vector<std::vector<uint32_t>*> to_sort;
        for (auto& kv : group) if (kv.first != 0) to_sort.push_back(&kv.second);
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < threads; ++t)
            th.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < to_sort.size();) std::stable_sort(to_sort[i]->begin(), to_sort[i]->end(), pair_greater); });
        for (auto& x : th) x.join();
    }
    for (const std::vector<int64_t>& bucket : buckets) {
        std::set<int64_t> sorted(bucket.begin(), bucket.end());
        for (int64_t b : sorted) {
            if (b == 0) continue;
            const std::vector<uint32_t>& g = group[b];
            if (reads_per_bc && 2 * g.size() >= reads_per_bc) continue;
            order.insert(order.end(), g.begin(), g.end());
            bci.push_back((int64_t)(2 * order.size()));
        }
    }
    // (= mergeBarcodedReadFiles' index, :251-288: 0, then the start of every barcode -- the first start is the number of
    // unbarcoded reads -- then the read count)

    // ---- encode: 2-bit bases and PQVec blocks per read, on all threads
    const size_t n_reads = 2 * order.size();
    std::vector<std::vector<uint8_t>> pk(n_reads), pq(n_reads);
    std::vector<uint32_t> lens(n_reads);
    {
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < threads; ++t)
            th.emplace_back([&] {
                std::vector<unsigned> cost; std::vector<PqBlock> blocks;
                for (size_t r; (r = next.fetch_add(256)) < n_reads;)
                    for (size_t k = r; k < std::min(n_reads, r + 256); ++k) {
                        const Fastq& f = (k & 1) ? f2 : f1;
                        const uint32_t i = order[k >> 1];
                        const uint8_t* b = f.bases.data() + f.off[i];
                        const uint32_t L = (uint32_t)(f.off[i + 1] - f.off[i]);
                        lens[k] = L;
                        pk[k].assign((L + 3) / 4, 0);
                        for (uint32_t j = 0; j < L; ++j) pk[k][j >> 2] |= (uint8_t)(b[j] << (2 * (j & 3)));
                        pq_encode(f.quals.data() + f.off[i], L, cost, blocks, &pq[k]);
                    }
            });
        for (auto& x : th) x.join();
    }
    std::vector<uint8_t> var_b, var_q;
    std::vector<uint64_t> off_b{0}, off_q{0};
    for (size_t k = 0; k < n_reads; ++k) {
        var_b.insert(var_b.end(), pk[k].begin(), pk[k].end()); off_b.push_back(var_b.size());
        var_q.insert(var_q.end(), pq[k].begin(), pq[k].end()); off_q.push_back(var_q.size());
    }
    try {
        const std::string head = a["OUT_HEAD"];
        const size_t slash = head.rfind('/');
        if (slash != std::string::npos) { std::string d = head.substr(0, slash); for (size_t i = 1; i <= d.size(); ++i) if (i == d.size() || d[i] == '/') mkdir(d.substr(0, i).c_str(), 0777); }
        feudal::write_fastb(head + ".fastb", var_b.data(), off_b, lens);
        feudal::write_qualp(head + ".qualp", var_q.data(), off_q);
        feudal::BinWriter w(head + ".bci"); w.vec(bci);
    } catch (const std::exception& e) { die(e.what()); }
    fprintf(stderr, "wrote %zu reads, %zu barcodes\n", n_reads, bci.size() - 2);
    return 0;
}
