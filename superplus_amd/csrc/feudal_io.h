// superplus_amd/csrc/feudal_io.h -- the reference's on-disk read formats, written from the format
// description in SURVEY.md 8b (not from the reference headers).  Host C++ only.
//
//   feudal file (feudal/FeudalControlBlock.h:159-165, feudal/FeudalFileWriter.cc:26-121):
//     24-byte header {u32 nElem, u8 flags(=1 file), u8 sizeofFixed, u8 sizeofX, u8 sizeofA,
//                     u64 varTabOffset = 24+varLen, u64 fixedOffset = varTabOffset + 8*(n+1)}
//     | var data | (n+1) x u64 ABSOLUTE offsets | fixed data
//   .fastb  var = ceil(len/4) bytes of 2-bit codes LSB-first, fixed = u32 len, header bytes 5-7 = 4,16,1
//   .qualp  var = PQVec block stream, no fixed data,                  header bytes 5-7 = 0,8,1
//   BINWRITE streams (feudal/BinaryStream.h:33-46,483-499): "BINWRITE" | payload; vec<T> = u64 n | elements
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace feudal {

struct Reads {
    std::vector<uint8_t> packed;     // .fastb var data
    std::vector<uint64_t> base_off;  // [n+1]
    std::vector<uint32_t> read_len;  // [n]
    std::vector<uint8_t> pq;         // .qualp var data
    std::vector<uint64_t> pq_off;    // [n+1]
    size_t size() const { return read_len.size(); }
};

inline std::vector<uint8_t> slurp(const std::string& path)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> b((size_t)n);
    if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); throw std::runtime_error("short read " + path); }
    fclose(f);
    return b;
}

struct Header { uint32_t n; uint8_t flags, szFixed, szX, szA; uint64_t varTab, fixedOff; };
static_assert(sizeof(Header) == 24, "feudal control block is 24 bytes");

inline void parse_feudal(const std::vector<uint8_t>& raw, const std::string& path, std::vector<uint8_t>* var,
                         std::vector<uint64_t>* off, std::vector<uint8_t>* fixed)
{
    if (raw.size() < 24) throw std::runtime_error(path + ": too short for a feudal file");
    Header h; memcpy(&h, raw.data(), 24);
    if ((h.flags & 3) != 1) throw std::runtime_error(path + ": not a single-file feudal file");
    if (h.varTab < 24 || h.fixedOff < h.varTab || h.fixedOff > raw.size() || (h.fixedOff - h.varTab) % 8 || h.fixedOff == h.varTab)
        throw std::runtime_error(path + ": inconsistent feudal control block");
    const uint64_t n = (h.fixedOff - h.varTab) / 8 - 1;
    if ((uint32_t)n != h.n) throw std::runtime_error(path + ": element count mismatch");
    off->resize(n + 1);
    memcpy(off->data(), raw.data() + h.varTab, 8 * (n + 1));
    for (uint64_t& o : *off) { if (o < 24 || o > h.varTab) throw std::runtime_error(path + ": offset out of range"); o -= 24; }
    var->assign(raw.begin() + 24, raw.begin() + h.varTab);
    if (fixed) fixed->assign(raw.begin() + h.fixedOff, raw.end());
}

inline void read_fastb(const std::string& path, std::vector<uint8_t>* packed, std::vector<uint64_t>* off, std::vector<uint32_t>* len)
{
    std::vector<uint8_t> fixed;
    parse_feudal(slurp(path), path, packed, off, &fixed);
    const size_t n = off->size() - 1;
    if (fixed.size() < 4 * n) throw std::runtime_error(path + ": fixed data too short");
    len->resize(n);
    memcpy(len->data(), fixed.data(), 4 * n);
    for (size_t i = 0; i < n; ++i)
        if ((*off)[i + 1] - (*off)[i] != ((*len)[i] + 3) / 4) throw std::runtime_error(path + ": read length disagrees with its byte count");
}

inline void read_qualp(const std::string& path, std::vector<uint8_t>* pq, std::vector<uint64_t>* off)
{ parse_feudal(slurp(path), path, pq, off, nullptr); }

inline void write_feudal(const std::string& path, const uint8_t* var, const std::vector<uint64_t>& off,
                         const void* fixed, size_t fixed_bytes, uint8_t szFixed, uint8_t szX, uint8_t szA)
{
    const uint64_t n = off.size() - 1, varLen = off[n];
    Header h{(uint32_t)n, 1, szFixed, szX, szA, 24 + varLen, 24 + varLen + 8 * (n + 1)};
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot create " + path);
    std::vector<uint64_t> abs(off);
    for (uint64_t& o : abs) o += 24;
    bool ok = fwrite(&h, 24, 1, f) == 1 && (varLen == 0 || fwrite(var, 1, varLen, f) == varLen) &&
              fwrite(abs.data(), 8, n + 1, f) == n + 1 && (fixed_bytes == 0 || fwrite(fixed, 1, fixed_bytes, f) == fixed_bytes);
    ok = (fclose(f) == 0) && ok;
    if (!ok) throw std::runtime_error("short write " + path);
}

inline void write_fastb(const std::string& path, const uint8_t* packed, const std::vector<uint64_t>& off, const std::vector<uint32_t>& len)
{ write_feudal(path, packed, off, len.data(), 4 * len.size(), 4, 16, 1); }
inline void write_qualp(const std::string& path, const uint8_t* pq, const std::vector<uint64_t>& off)
{ write_feudal(path, pq, off, nullptr, 0, 0, 8, 1); }

// ---- BINWRITE streams ----
struct BinWriter {
    FILE* f;
    explicit BinWriter(const std::string& path) : f(fopen(path.c_str(), "wb"))
    { if (!f) throw std::runtime_error("cannot create " + path); raw("BINWRITE", 8); }
    ~BinWriter() { if (f) fclose(f); }
    void raw(const void* p, size_t n) { if (n && fwrite(p, 1, n, f) != n) throw std::runtime_error("short write"); }
    template <class T> void pod(const T& v) { raw(&v, sizeof v); }
    template <class T> void vec(const std::vector<T>& v) { pod<uint64_t>(v.size()); raw(v.data(), sizeof(T) * v.size()); }
    // FeudalString::writeBinary (feudal/FeudalString.h:487-491): u32 length including the NUL, then the bytes + NUL
    void str(const std::string& s) { pod<uint32_t>((uint32_t)s.size() + 1); raw(s.c_str(), s.size() + 1); }
};

inline std::vector<int64_t> read_bci(const std::string& path)
{
    std::vector<uint8_t> raw = slurp(path);
    if (raw.size() < 16 || memcmp(raw.data(), "BINWRITE", 8)) throw std::runtime_error(path + ": missing BINWRITE magic");
    uint64_t n; memcpy(&n, raw.data() + 8, 8);
    if (raw.size() < 16 + 8 * n) throw std::runtime_error(path + ": truncated");
    std::vector<int64_t> v(n);
    memcpy(v.data(), raw.data() + 16, 8 * n);
    return v;
}

} // namespace feudal
