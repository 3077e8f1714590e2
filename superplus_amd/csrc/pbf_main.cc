// superplus_amd/csrc/pbf_main.cc -- `ParseBarcodedFastqs`: stLFR fastq.gz pairs -> barcode-sorted
// OUT_HEAD.{fastb,qualp,bci}, the files DF reads (runall.sh:125).  Host-only ingest, written from the behaviour of
// 10X/ParseBarcodedFastqs.cc:306-539 + mergeBarcodedReadFiles :222-304, not from its structure:
//
//   the reference reads both gz files once per barcode BUCKET (NUM_BUCKETS passes over the input), keeps each
//   barcode's pairs in a list it insertion-sorts (quadratic per barcode), writes one temporary file set per bucket
//   and merges them; here the files are read ONCE (one thread per file) when their decompressed reads fit the memory
//   allowed, pairs are grouped by barcode, each group is stable-sorted, qualities are encoded on all threads and the three
//   files are written directly.  When they do not fit, the first pass keeps the barcodes and lengths only, and the files
//   are read again once per GROUP of buckets -- as many consecutive buckets as the memory holds -- each group appended
//   to the output as it is done: the reference's bound on memory with (usually far) fewer passes than NUM_BUCKETS.
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
#include "../../include/dfk.h"
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

std::vector<std::string> g_partial;          // output files begun and not finished: a run that gives up takes them along
[[noreturn]] void die(const std::string& m)
{
    fprintf(stderr, "ParseBarcodedFastqs: %s\n", m.c_str());
    for (const std::string& p : g_partial) remove(p.c_str());
    exit(1);
}

std::vector<std::string> parse_set(std::string s)
{
    s.erase(std::remove(s.begin(), s.end(), '{'), s.end());
    s.erase(std::remove(s.begin(), s.end(), '}'), s.end());
    std::vector<std::string> out; std::stringstream ss(s); std::string tok;
    while (std::getline(ss, tok, ',')) if (!tok.empty()) out.push_back(tok);
    return out;
}

struct Fastq {                               // one file: per read its barcode; base codes and qualities of the reads that are HELD
    std::vector<int64_t> bc;                 // of every read (first pass)
    std::vector<uint16_t> len;               // of every read (first pass)
    std::vector<uint8_t> bases, quals;       // concatenated, held reads only
    std::vector<uint64_t> off{0};            // [held + 1] into bases / quals
    std::string error;
    uint64_t counted = 0;                    // bases already added to g_held
    bool raw = false;                        // device path: bases and quals keep the file's characters (the device turns them into codes)
    bool spilled = false;                    // the first pass gave the bases up (the set does not fit): bc and len are complete, the rest empty
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

// Memory: the reads that are held cost a byte per base and a byte per quality, and then their encoded form twice while a
// group is written -- about 3.5 bytes per base plus ~100 per read.  g_mem_limit (MAX_MEM_GB, or the machine's memory) is
// what all of it may take: a first pass that outgrows it drops the bases (g_spill) and only scans on, and the buckets are
// then done in groups that fit (the reference bounds its own use by re-reading the inputs once per bucket,
// 10X/ParseBarcodedFastqs.cc:434-449).  Only a SINGLE bucket too large for the memory ends the program, with a message
// that says so rather than with the kernel's OOM killer.
std::atomic<uint64_t> g_held{0};
std::atomic<bool> g_spill{false};
uint64_t g_mem_limit = 0;
bool over_budget(uint64_t bases, uint64_t reads) { return g_mem_limit && 7 * bases / 2 + 100 * reads > g_mem_limit; }
std::string budget_message(uint64_t bases, uint64_t reads)
{
    return "one barcode bucket needs about " + std::to_string((7 * bases / 2 + 100 * reads) >> 30) + " GiB (3.5 bytes per base + 100 per read: a bucket's decompressed reads are held in "
           "memory), more than the " + std::to_string(g_mem_limit >> 30) + " GiB allowed (MAX_MEM_GB, or the machine's memory); raise NUM_BUCKETS (256 at most) or the memory";
}

// want = nullptr: the first pass -- every read's barcode and length; its bases and qualities too until the budget says no.
// want != nullptr: a group's pass -- bases and qualities of the reads i (numbered as they stand in this file) with (*want)[i];
// nothing else is stored.
void read_fastq(const std::string& path, Fastq* f, const std::vector<uint8_t>* want)
{
    gzFile g = gzopen(path.c_str(), "rb");
    if (!g) { f->error = "cannot open " + path; return; }
    gzbuffer(g, 1 << 20);
    std::vector<char> line(1 << 16);
    auto get = [&]() -> bool { return gzgets(g, line.data(), (int)line.size()) != nullptr; };
    uint64_t i = 0;                                                                    // reads seen
    while (get()) {
        if (line[0] != '@') { f->error = "out of sync reading line: " + path + ": " + line.data(); break; }
        int64_t bc = 0;
        if (!want && !barcode_of(line.data(), &bc)) { f->error = "cannot parse the barcode of " + std::string(line.data()); break; }
        if (!get()) { f->error = "truncated record in " + path; break; }
        const bool hold = want ? (i < want->size() && (*want)[i]) : !f->spilled;
        size_t nb = 0;
        if (f->raw) {                                                                   // the line as it is: the device looks at the characters
            nb = strcspn(line.data(), "\r\n");
            if (hold) f->bases.insert(f->bases.end(), (const uint8_t*)line.data(), (const uint8_t*)line.data() + nb);
        } else
        for (const char* p = line.data(); *p && *p != '\n' && *p != '\r'; ++p, ++nb) {
            uint8_t v;
            switch (*p) { case 'A': case 'a': case 'N': case 'n': v = 0; break; case 'C': case 'c': v = 1; break;
                          case 'G': case 'g': v = 2; break; case 'T': case 't': v = 3; break;
                          default: f->error = std::string("unexpected base '") + *p + "' in " + path; v = 0; }
            if (hold) f->bases.push_back(v);
        }
        if (!get() || !get()) { f->error = "truncated record in " + path; break; }       // '+' line, then the qualities
        size_t nq = 0;
        if (f->raw) {
            for (const char* p = line.data(); *p; ++p) if (*p != '\n' && *p != '\r') { if (hold) f->quals.push_back((uint8_t)*p); ++nq; }
        } else
        for (const char* p = line.data(); *p; ++p) if (*p != '\n' && *p != '\r') { if (hold) f->quals.push_back((uint8_t)(*p - 33)); ++nq; }
        if (nq != nb) { f->error = "a read of " + path + " has " + std::to_string(nq) + " qualities for " + std::to_string(nb) + " bases"; break; }
        if (!want) {
            if (nb > 65535) { f->error = "a read of " + path + " has " + std::to_string(nb) + " bases (65535 at most)"; break; }
            f->bc.push_back(bc); f->len.push_back((uint16_t)nb);
        }
        if (hold) f->off.push_back(f->bases.size());
        ++i;
        if (!f->error.empty()) break;
        if (!want && (i & 0xFFFF) == 0) {                                              // (both reader threads add to one total)
            if (!f->spilled) {
                const uint64_t mine = f->bases.size();
                const uint64_t all = g_held.fetch_add(mine - f->counted) + (mine - f->counted);
                f->counted = mine;
                if (over_budget(all, 0)) g_spill = true;
            }
            if (g_spill && !f->spilled) {                                              // the set does not fit: scan on, hold nothing
                f->spilled = true;
                std::vector<uint8_t>().swap(f->bases); std::vector<uint8_t>().swap(f->quals); std::vector<uint64_t>(1, 0).swap(f->off);
            }
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
                                            {"NUM_THREADS", "0"}, {"MAX_MEM_GB", "0"}, {"MERGE_HEADS", ""}, {"HOST", "False"}, {"DEVICE", "0"}};
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
    // ---- where the data-parallel half runs.  The product is the device (dfk_pbf_run: pair order, 2-bit packing, PQVec encoding on
    //      the MI355X) and a run without one fails; HOST=True is this file's own host code for all of it, chosen explicitly --
    //      what the byte-for-byte comparisons with the reference's binary run on in the build container, which has no GPU.
    const bool host_only = a["HOST"] == "True" || a["HOST"] == "true" || a["HOST"] == "1";
    dfk_ctx* ctx = nullptr;
    if (!host_only) {
        dfk_config cfg{};
        cfg.abi_version = DFK_ABI_VERSION; cfg.K = 48; cfg.min_qual = 7; cfg.min_freq = 3; cfg.min_bc = 2; cfg.device = atoi(a["DEVICE"].c_str());
        if (dfk_create(&cfg, &ctx)) die(std::string(dfk_last_error()) + " (HOST=True runs the packing and encoding on the host)");
    }
    double dev_ms[3] = {0, 0, 0};
    // ---- both files, one thread each
    Fastq f1, f2;
    f1.raw = f2.raw = !host_only;
    { std::thread t(read_fastq, fq[1], &f2, nullptr); read_fastq(fq[0], &f1, nullptr); t.join(); }
    if (!f1.error.empty()) die(f1.error);
    if (!f2.error.empty()) die(f2.error);
    if (f1.bc.size() != f2.bc.size()) die("something not match with pair file: " + fq[0] + " or " + fq[1]);
    const size_t n_pairs = f1.bc.size();
    for (size_t i = 0; i < n_pairs; ++i) if (f1.bc[i] != f2.bc[i]) die("something not match with pair file: " + fq[0] + " or " + fq[1]);
    fprintf(stderr, "total reads: %zu\n", 2 * n_pairs);
    // (one reader may have reached the budget's check after the other had finished: both or neither hold their reads)
    const bool spilled = f1.spilled || f2.spilled || over_budget(f1.bases.size() + f2.bases.size(), 2 * n_pairs);
    if (spilled) for (Fastq* f : {&f1, &f2}) { std::vector<uint8_t>().swap(f->bases); std::vector<uint8_t>().swap(f->quals); std::vector<uint64_t>(1, 0).swap(f->off); }
    const std::vector<int64_t>& bc = f1.bc;
    std::vector<int64_t>().swap(f2.bc);

    // ---- buckets of barcodes (:311-336): the distinct barcodes in the container's iteration order, cut into equal runs
    std::unordered_set<int64_t> bc_set;
    for (size_t i = 0; i < n_pairs; ++i) bc_set.emplace(bc[i]);
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

    // ---- the output, appended to as the units are done: the unbarcoded pairs in file order (unit 0), then one unit per bucket
    const std::string head = a["OUT_HEAD"];
    { const size_t slash = head.rfind('/');
      if (slash != std::string::npos) { std::string d = head.substr(0, slash); for (size_t i = 1; i <= d.size(); ++i) if (i == d.size() || d[i] == '/') mkdir(d.substr(0, i).c_str(), 0777); } }
    struct Appender {                                                           // a feudal file whose variable data comes in pieces
        FILE* f = nullptr; std::string path; std::vector<uint64_t> off{24};
        void open(const std::string& p) { path = p; f = fopen(p.c_str(), "wb"); g_partial.push_back(p); const char z[24] = {0}; if (!f || fwrite(z, 1, 24, f) != 24) die("cannot create " + p); }
        void add(const std::vector<uint8_t>& v) { if (!v.empty() && fwrite(v.data(), 1, v.size(), f) != v.size()) die("short write " + path); off.push_back(off.back() + v.size()); }
        void add_many(const uint8_t* var, const uint64_t* rel, uint64_t n)       // n elements whose bytes lie one behind the other in var, at rel[0..n]
        {
            if (rel[n] && fwrite(var, 1, rel[n], f) != rel[n]) die("short write " + path);
            const uint64_t base = off.back();
            for (uint64_t i = 1; i <= n; ++i) off.push_back(base + rel[i]);
        }
        void finish(const void* fixed, size_t fixed_bytes, uint8_t szFixed, uint8_t szX, uint8_t szA)
        {
            const uint64_t n = off.size() - 1;
            feudal::Header h{(uint32_t)n, 1, szFixed, szX, szA, off[n], off[n] + 8 * (n + 1)};
            bool ok = fwrite(off.data(), 8, n + 1, f) == n + 1 && (fixed_bytes == 0 || fwrite(fixed, 1, fixed_bytes, f) == fixed_bytes);
            ok = ok && fseek(f, 0, SEEK_SET) == 0 && fwrite(&h, 24, 1, f) == 1;
            ok = (fclose(f) == 0) && ok; f = nullptr;
            if (!ok) die("short write " + path);
        }
    } out_b, out_q;
    out_b.open(head + ".fastb"); out_q.open(head + ".qualp");
    std::vector<uint32_t> lens;
    std::vector<int64_t> bci{0};
    size_t n_barcodes = 0;
    // (= mergeBarcodedReadFiles' index, :251-288: 0, then the start of every barcode -- the first start is the number of
    // unbarcoded reads -- then the read count)

    // The reads held in g1 / g2 are those of the pairs `held` names (ascending; nullptr = all pairs).  Writes the units [u0, u1):
    // unit 0 = the unbarcoded pairs among them, unit u = bucket u - 1.
    auto emit = [&](const Fastq& g1, const Fastq& g2, const std::vector<uint32_t>* held, size_t u0, size_t u1, bool first_of_unit0, bool last_of_unit0) {
        const size_t m = held ? held->size() : n_pairs;
        auto pair_no = [&](size_t k) { return held ? (size_t)(*held)[k] : k; };
        if (ctx) {
            // ---- the device path: this side decides only WHERE every barcode stands (unit 0 = the unbarcoded pairs in file order, then
            //      the buckets in order, inside a bucket the barcodes ascending, :346) and how many pairs each has (the index); the
            //      order inside a barcode, the 2-bit bases and the PQVec streams come from dfk_pbf_run
            std::unordered_map<int64_t, uint32_t> count;
            for (size_t k = 0; k < m; ++k) ++count[bc[pair_no(k)]];
            std::unordered_map<int64_t, uint32_t> place;
            const uint64_t reads_before = out_b.off.size() - 1;
            uint64_t pairs_so_far = 0;
            if (u0 == 0) {
                if (count.count(0)) { place[0] = 0; pairs_so_far += count[0]; }
                if (last_of_unit0) bci.push_back((int64_t)(reads_before + 2 * pairs_so_far));
            }
            uint32_t next_place = 1;
            for (size_t u = std::max<size_t>(u0, 1); u < u1; ++u) {
                const std::vector<int64_t>& bucket = buckets[u - 1];
                std::set<int64_t> sorted(bucket.begin(), bucket.end());
                for (int64_t b : sorted) {
                    if (b == 0) continue;
                    const uint32_t n_b = count.count(b) ? count[b] : 0;
                    if (reads_per_bc && 2 * (size_t)n_b >= reads_per_bc) continue;
                    place[b] = next_place++;
                    pairs_so_far += n_b;
                    bci.push_back((int64_t)(reads_before + 2 * pairs_so_far));
                    ++n_barcodes;
                }
            }
            std::vector<uint32_t> rank(m);
            for (size_t k = 0; k < m; ++k) { const auto it = place.find(bc[pair_no(k)]); rank[k] = it == place.end() ? DFK_PBF_DROP : it->second; }
            dfk_pbf_input in{{g1.bases.data(), g2.bases.data()}, {g1.quals.data(), g2.quals.data()}, {g1.off.data(), g2.off.data()}, rank.data(), (uint64_t)m};
            dfk_pbf* res = nullptr;
            if (dfk_pbf_run(ctx, &in, &res)) die(dfk_last_error());
            dfk_pbf_output o{};
            dfk_pbf_result(res, &o);
            if (o.n_pairs != pairs_so_far) die("the device wrote " + std::to_string(o.n_pairs) + " pairs where the barcode index counts " + std::to_string(pairs_so_far));
            lens.insert(lens.end(), o.read_len, o.read_len + 2 * o.n_pairs);
            out_b.add_many(o.fastb_var, o.fastb_off, 2 * o.n_pairs);
            out_q.add_many(o.qualp_var, o.qualp_off, 2 * o.n_pairs);
            dev_ms[0] += o.ms_sort; dev_ms[1] += o.ms_encode; dev_ms[2] += o.ms_total;
            dfk_pbf_free(res);
            return;
        }
        // ---- pairs grouped by barcode, in file order
        std::unordered_map<int64_t, std::vector<uint32_t>> group;
        for (size_t k = 0; k < m; ++k) group[bc[pair_no(k)]].push_back((uint32_t)k);
        auto seq = [](const Fastq& f, uint32_t i) { return std::make_pair(f.bases.data() + f.off[i], f.bases.data() + f.off[i + 1]); };
        auto pair_greater = [&](uint32_t x, uint32_t y) {              // (read 1, read 2) of x above those of y, as base-code sequences
            const auto x1 = seq(g1, x), y1 = seq(g1, y);
            if (std::lexicographical_compare(y1.first, y1.second, x1.first, x1.second)) return true;
            if (std::lexicographical_compare(x1.first, x1.second, y1.first, y1.second)) return false;
            const auto x2 = seq(g2, x), y2 = seq(g2, y);
            return std::lexicographical_compare(y2.first, y2.second, x2.first, x2.second);
        };
        // ---- output order of the pairs and the barcode index
        std::vector<uint32_t> order; order.reserve(m);
        const uint64_t reads_before = out_b.off.size() - 1;
        if (u0 == 0) {
            (void)first_of_unit0;
            if (group.count(0)) order = group[0];
            if (last_of_unit0) bci.push_back((int64_t)(reads_before + 2 * order.size()));
        }
        {
            std::vector<std::vector<uint32_t>*> to_sort;
            for (auto& kv : group) if (kv.first != 0) to_sort.push_back(&kv.second);
            std::atomic<size_t> next{0};
            std::vector<std::thread> th;
            for (unsigned t = 0; t < threads; ++t)
                th.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < to_sort.size();) std::stable_sort(to_sort[i]->begin(), to_sort[i]->end(), pair_greater); });
            for (auto& x : th) x.join();
        }
        for (size_t u = std::max<size_t>(u0, 1); u < u1; ++u) {
            const std::vector<int64_t>& bucket = buckets[u - 1];
            std::set<int64_t> sorted(bucket.begin(), bucket.end());
            for (int64_t b : sorted) {
                if (b == 0) continue;
                const std::vector<uint32_t>& g = group[b];
                if (reads_per_bc && 2 * g.size() >= reads_per_bc) continue;
                order.insert(order.end(), g.begin(), g.end());
                bci.push_back((int64_t)(reads_before + 2 * order.size()));
                ++n_barcodes;
            }
        }
        // ---- encode: 2-bit bases and PQVec blocks per read, on all threads
        const size_t n_reads = 2 * order.size();
        std::vector<std::vector<uint8_t>> pk(n_reads), pq(n_reads);
        const size_t lens0 = lens.size();
        lens.resize(lens0 + n_reads);
        {
            std::atomic<size_t> next{0};
            std::vector<std::thread> th;
            for (unsigned t = 0; t < threads; ++t)
                th.emplace_back([&] {
                    std::vector<unsigned> cost; std::vector<PqBlock> blocks;
                    for (size_t r; (r = next.fetch_add(256)) < n_reads;)
                        for (size_t k = r; k < std::min(n_reads, r + 256); ++k) {
                            const Fastq& f = (k & 1) ? g2 : g1;
                            const uint32_t i = order[k >> 1];
                            const uint8_t* b = f.bases.data() + f.off[i];
                            const uint32_t L = (uint32_t)(f.off[i + 1] - f.off[i]);
                            lens[lens0 + k] = L;
                            pk[k].assign((L + 3) / 4, 0);
                            for (uint32_t j = 0; j < L; ++j) pk[k][j >> 2] |= (uint8_t)(b[j] << (2 * (j & 3)));
                            pq_encode(f.quals.data() + f.off[i], L, cost, blocks, &pq[k]);
                        }
                });
            for (auto& x : th) x.join();
        }
        for (size_t k = 0; k < n_reads; ++k) { out_b.add(pk[k]); out_q.add(pq[k]); }
    };

    if (!spilled) emit(f1, f2, nullptr, 0, buckets.size() + 1, true, true);
    else {
        // ---- the set does not fit: groups of consecutive units that do, the inputs read again for each (held reads only)
        std::vector<uint16_t> len1, len2; len1.swap(f1.len); len2.swap(f2.len);
        // what the tables kept for the whole run take (barcodes, lengths, the flags of a pass, the output's offsets and lengths)
        const uint64_t tables = (uint64_t)n_pairs * (8 + 2 + 2 + 1 + 1 + 2 * 20);
        if (g_mem_limit && tables >= g_mem_limit) die("the tables of " + std::to_string(n_pairs) + " pairs alone take " + std::to_string(tables >> 30) + " GiB, more than MAX_MEM_GB (or the machine's memory) allows");
        std::unordered_map<int64_t, uint32_t> unit_of;                          // barcode -> unit
        for (size_t u = 0; u < buckets.size(); ++u) for (int64_t b : buckets[u]) unit_of[b] = (uint32_t)(u + 1);
        unit_of[0] = 0;
        std::vector<uint64_t> unit_bases(buckets.size() + 1, 0), unit_pairs(buckets.size() + 1, 0);
        for (size_t i = 0; i < n_pairs; ++i) { const uint32_t u = unit_of[bc[i]]; unit_bases[u] += (uint64_t)len1[i] + len2[i]; ++unit_pairs[u]; }
        auto fits = [&](uint64_t bases, uint64_t pairs) { return !g_mem_limit || 7 * bases / 2 + 200 * pairs + tables <= g_mem_limit; };
        auto pass = [&](const std::vector<uint8_t>& want, const std::vector<uint32_t>& held, size_t u0, size_t u1, bool first0, bool last0) {
            Fastq g1, g2;
            g1.raw = g2.raw = ctx != nullptr;
            { std::thread t(read_fastq, fq[1], &g2, &want); read_fastq(fq[0], &g1, &want); t.join(); }
            if (!g1.error.empty()) die(g1.error);
            if (!g2.error.empty()) die(g2.error);
            if (g1.off.size() - 1 != held.size() || g2.off.size() - 1 != held.size()) die("the input changed between two passes over it");
            emit(g1, g2, &held, u0, u1, first0, last0);
        };
        size_t passes = 0;
        // unit 0 keeps the file's order: it may be cut anywhere
        {
            size_t i = 0;
            bool first = true;
            const bool any0 = unit_pairs[0] != 0;
            if (!any0) bci.push_back(0);
            while (any0 && i < n_pairs) {
                std::vector<uint8_t> want(n_pairs, 0); std::vector<uint32_t> held;
                uint64_t bases = 0;
                for (; i < n_pairs; ++i) {
                    if (bc[i] != 0) continue;
                    const uint64_t add = (uint64_t)len1[i] + len2[i];
                    if (!held.empty() && !fits(bases + add, held.size() + 1)) break;
                    bases += add; want[i] = 1; held.push_back((uint32_t)i);
                }
                if (held.empty()) break;
                bool more = false;
                for (size_t j = i; j < n_pairs && !more; ++j) more = bc[j] == 0;
                pass(want, held, 0, 1, first, !more); ++passes;
                first = false;
            }
        }
        for (size_t u = 1; u <= buckets.size();) {
            size_t v = u; uint64_t bases = 0, pairs = 0;
            while (v <= buckets.size() && (v == u || fits(bases + unit_bases[v], pairs + unit_pairs[v]))) { bases += unit_bases[v]; pairs += unit_pairs[v]; ++v; }
            if (!fits(bases, pairs)) die(budget_message(bases, 2 * pairs));
            std::vector<uint8_t> want(n_pairs, 0); std::vector<uint32_t> held;
            for (size_t i = 0; i < n_pairs; ++i) { const uint32_t w = unit_of[bc[i]]; if (w >= u && w < v) { want[i] = 1; held.push_back((uint32_t)i); } }
            pass(want, held, u, v, false, false); ++passes;
            u = v;
        }
        fprintf(stderr, "the reads do not fit the memory allowed: %zu more passes over the input\n", passes);
    }
    const size_t n_reads = out_b.off.size() - 1;
    try {
        out_b.finish(lens.data(), 4 * lens.size(), 4, 16, 1);
        out_q.finish(nullptr, 0, 0, 8, 1);
        feudal::BinWriter w(head + ".bci"); w.vec(bci);
    } catch (const std::exception& e) { die(e.what()); }
    g_partial.clear();
    if (ctx) { fprintf(stderr, "device: pair order %.1f ms, packing + PQVec %.1f ms, with transfers %.1f ms\n", dev_ms[0], dev_ms[1], dev_ms[2]); dfk_destroy(ctx); }
    fprintf(stderr, "wrote %zu reads, %zu barcodes\n", n_reads, n_barcodes);
    return 0;
}
