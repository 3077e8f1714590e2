"""Python view of the reference's on-disk read formats (host-side harness only).

The product's own reader/writer is C++ (superplus_amd/csrc/feudal_io.*); this module
exists so tests and bench.py can build and inspect inputs with numpy.

Formats (reference paths relative to lib/assembly/src):
  feudal file  feudal/FeudalControlBlock.h:159-165, feudal/FeudalFileWriter.cc:26-121
      24-B header {u32 nElem, u8 flags(=1 file), u8 sizeofFixed, u8 sizeofX, u8 sizeofA,
                   u64 varTabOffset = 24+varLen, u64 fixedOffset = varTabOffset + 8*(n+1)}
      | var data | (n+1) x u64 ABSOLUTE offsets | fixed data
  .fastb       var = ceil(len/4) bytes, 2-bit codes LSB-first (feudal/FieldVec.h:766-770),
               fixed = u32 len; header bytes 5-7 = (4, 16, 1)
  .qualp       var = PQVec bytes (feudal/PQVec.h:158-161), no fixed; header bytes 5-7 = (0, 8, 1)
  .bci         "BINWRITE" | u64 n | n x i64   (feudal/BinaryStream.h:33-46,483-499)
"""
import struct

import numpy as np

_HDR = struct.Struct("<IBBBBQQ")


def header(n, sizeof_fixed, sizeof_x, sizeof_a, var_tab, fixed_off):
    """The 24-byte control block of a feudal file."""
    return _HDR.pack(n & 0xFFFFFFFF, 1, sizeof_fixed, sizeof_x, sizeof_a, var_tab, fixed_off)


def _write_feudal(path, var_bytes, var_off, fixed_bytes, sizeof_fixed, sizeof_x, sizeof_a):
    n = len(var_off) - 1
    var_len = int(var_off[-1])
    var_tab = 24 + var_len
    fixed_off = var_tab + 8 * (n + 1)
    with open(path, "wb") as f:
        f.write(_HDR.pack(n & 0xFFFFFFFF, 1, sizeof_fixed, sizeof_x, sizeof_a, var_tab, fixed_off))
        f.write(np.ascontiguousarray(var_bytes, dtype=np.uint8)[:var_len].tobytes())
        f.write((np.asarray(var_off, dtype=np.uint64) + np.uint64(24)).tobytes())
        if fixed_bytes is not None:
            f.write(fixed_bytes)


def _read_feudal(path):
    raw = np.fromfile(path, dtype=np.uint8)
    n_mod, flags, sz_fixed, sz_x, sz_a, var_tab, fixed_off = _HDR.unpack(raw[:24].tobytes())
    if flags & 3 != 1:
        raise ValueError(f"{path}: not a single-file feudal file (flags={flags})")
    n = (fixed_off - var_tab) // 8 - 1
    if n & 0xFFFFFFFF != n_mod:
        raise ValueError(f"{path}: element count mismatch")
    offs = raw[var_tab:fixed_off].view(np.uint64).astype(np.uint64) - np.uint64(24)
    var = raw[24:var_tab]
    fixed = raw[fixed_off:]
    return var, offs, fixed, (sz_fixed, sz_x, sz_a)


def write_fastb(path, packed, base_off, read_len):
    read_len = np.asarray(read_len, dtype=np.uint32)
    _write_feudal(path, packed, base_off, read_len.tobytes(), 4, 16, 1)


def read_fastb(path):
    """-> (packed u8[], base_off u64[n+1], read_len u32[n])"""
    var, offs, fixed, _ = _read_feudal(path)
    n = len(offs) - 1
    read_len = fixed[: 4 * n].view(np.uint32).copy()
    return var.copy(), offs, read_len


def write_qualp(path, pq_bytes, pq_off):
    _write_feudal(path, pq_bytes, pq_off, None, 0, 8, 1)


def read_qualp(path):
    """-> (pq_bytes u8[], pq_off u64[n+1])"""
    var, offs, _, _ = _read_feudal(path)
    return var.copy(), offs


def write_bci(path, bci):
    bci = np.asarray(bci, dtype=np.int64)
    with open(path, "wb") as f:
        f.write(b"BINWRITE")
        f.write(struct.pack("<Q", len(bci)))
        f.write(bci.tobytes())


def read_bci(path):
    raw = open(path, "rb").read()
    if raw[:8] != b"BINWRITE":
        raise ValueError(f"{path}: missing BINWRITE magic")
    (n,) = struct.unpack("<Q", raw[8:16])
    return np.frombuffer(raw, dtype=np.int64, count=n, offset=16).copy()


def bci_to_bc(bci, n_reads):
    """DF.cc:447-452: expand the barcode index to one int32 barcode id per read (0 = unbarcoded)."""
    bci = np.asarray(bci, dtype=np.int64)
    bc = np.zeros(n_reads, dtype=np.int32)
    if len(bci) >= 2:
        counts = np.diff(bci)
        bc[: int(bci[-1])] = np.repeat(np.arange(len(counts), dtype=np.int32), counts)
    return bc


def pack_bases(codes):
    """codes: u8[n, L] base codes 0..3 -> packed u8[n, ceil(L/4)] (LSB-first within byte)."""
    codes = np.asarray(codes, dtype=np.uint8)
    n, L = codes.shape
    pad = (-L) % 4
    if pad:
        codes = np.concatenate([codes, np.zeros((n, pad), dtype=np.uint8)], axis=1)
    c = codes.reshape(n, -1, 4)
    return (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8)


def pq_encode(q):
    """A valid PQVec encoding of one quality vector (block layout of feudal/PQVec.cc:87-127).
    One block per run of <=255 values; width from the run's range.  Slow; for small tests."""
    q = [int(x) for x in q]
    out = bytearray()
    i = 0
    while i < len(q):
        j = i + 1
        while j < len(q) and j - i < 255 and q[j] == q[i]:
            j += 1
        if j - i < 4:  # not a constant run: take a mixed block up to the next long constant run
            j = i + 1
            while j < len(q) and j - i < 255:
                if j + 4 <= len(q) and len(set(q[j : j + 4])) == 1 and q[j] != q[j - 1]:
                    break
                j += 1
        blk = q[i:j]
        mn, mx = min(blk), max(blk)
        nbits = (mx - mn).bit_length()
        bits = nbits | (mn << 3)
        pos = 9
        for v in blk:
            bits |= (v - mn) << pos
            pos += nbits
        out.append(len(blk))
        out += bits.to_bytes((pos + 7) // 8, "little")
        i = j
    out.append(0)
    return bytes(out)
