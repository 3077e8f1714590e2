"""ctypes view of oracle/libdfk_oracle.so -- TEST INFRASTRUCTURE.

Import only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ENTRY_DTYPE = np.dtype([("w0", "<u8"), ("w1", "<u8"), ("edge_id", "<u4"), ("count_ctx", "<u4"),
                        ("bc", "<i4"), ("pad", "<u4")])
INST_DTYPE = np.dtype([("w0", "<u8"), ("w1", "<u8"), ("bc", "<i4"), ("ctx", "<u4")])


class _Result(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("good_len", C.POINTER(C.c_uint32)), ("n_inst", C.c_uint64),
                ("n_distinct", C.c_uint64), ("n_solid", C.c_uint64), ("solid_pre", C.c_void_p),
                ("solid", C.c_void_p), ("n_bins", C.c_uint64), ("hist", C.POINTER(C.c_int64)),
                ("t_trim", C.c_double), ("t_kmerize", C.c_double), ("t_count", C.c_double), ("t_adj", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libdfk_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libdfk_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.dfko_run.restype = C.POINTER(_Result)
        L.dfko_run_goodlen.restype = C.POINTER(_Result)
        L.dfko_fnv1a16.restype = C.c_uint64
        L.dfko_pq_decode.restype = C.c_int64
        L.dfko_pq_encode.restype = C.c_uint64
        L.dfko_good_len.restype = C.c_uint32
        L.dfko_kmerize.restype = C.c_uint64
        L.dfko_spectrum_json.restype = C.c_uint64
        L.dfko_ctx_rc.restype = C.c_uint8
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _harvest(rp):
    if not rp:
        raise RuntimeError("oracle: malformed input (PQVec length mismatch)")
    r = rp.contents
    ns = int(r.n_solid)
    def ent(ptr):
        if ns == 0:
            return np.zeros(0, dtype=ENTRY_DTYPE)
        return np.frombuffer(C.string_at(ptr, 32 * ns), dtype=ENTRY_DTYPE).copy()
    out = dict(
        good_len=np.ctypeslib.as_array(r.good_len, shape=(int(r.n_reads),)).copy() if r.n_reads else np.zeros(0, np.uint32),
        n_inst=int(r.n_inst), n_distinct=int(r.n_distinct), n_solid=ns,
        solid_pre=ent(r.solid_pre), solid=ent(r.solid),
        hist=np.ctypeslib.as_array(r.hist, shape=(int(r.n_bins),)).copy() if r.n_bins else np.zeros(0, np.int64),
        t_trim=r.t_trim, t_kmerize=r.t_kmerize, t_count=r.t_count, t_adj=r.t_adj)
    lib().dfko_free(rp)
    return out


def run(packed, base_off, read_len, pq_bytes, pq_off, bc, K=48, min_qual=7, min_freq=3, min_bc=2,
        ign_bc_below=0, threads=0):
    packed = np.ascontiguousarray(packed, np.uint8); base_off = np.ascontiguousarray(base_off, np.uint64)
    read_len = np.ascontiguousarray(read_len, np.uint32); pq_bytes = np.ascontiguousarray(pq_bytes, np.uint8)
    pq_off = np.ascontiguousarray(pq_off, np.uint64)
    bc = None if bc is None else np.ascontiguousarray(bc, np.int32)
    n = len(read_len)
    rp = lib().dfko_run(_p(packed), _p(base_off), _p(read_len), _p(pq_bytes), _p(pq_off), _p(bc),
                        C.c_uint64(n), C.c_uint(K), C.c_uint(min_qual), C.c_uint(min_freq), C.c_uint(min_bc),
                        C.c_int64(ign_bc_below), C.c_int(threads))
    return _harvest(rp)


def run_goodlen(packed, base_off, good_len, bc, K=48, min_freq=3, min_bc=2, ign_bc_below=0, threads=0):
    packed = np.ascontiguousarray(packed, np.uint8); base_off = np.ascontiguousarray(base_off, np.uint64)
    good_len = np.ascontiguousarray(good_len, np.uint32)
    bc = None if bc is None else np.ascontiguousarray(bc, np.int32)
    rp = lib().dfko_run_goodlen(_p(packed), _p(base_off), _p(good_len), _p(bc), C.c_uint64(len(good_len)),
                                C.c_uint(K), C.c_uint(min_freq), C.c_uint(min_bc), C.c_int64(ign_bc_below),
                                C.c_int(threads))
    return _harvest(rp)


def kmerize(packed, base_off, good_len, bc, K=48, ign_bc_below=0):
    packed = np.ascontiguousarray(packed, np.uint8); base_off = np.ascontiguousarray(base_off, np.uint64)
    good_len = np.ascontiguousarray(good_len, np.uint32)
    bc = None if bc is None else np.ascontiguousarray(bc, np.int32)
    n = lib().dfko_kmerize(_p(packed), _p(base_off), _p(good_len), _p(bc), C.c_int64(ign_bc_below),
                           C.c_uint64(len(good_len)), C.c_uint(K), None, C.c_uint64(0))
    out = np.zeros(int(n), dtype=INST_DTYPE)
    lib().dfko_kmerize(_p(packed), _p(base_off), _p(good_len), _p(bc), C.c_int64(ign_bc_below),
                       C.c_uint64(len(good_len)), C.c_uint(K), _p(out), C.c_uint64(n))
    return out


def pq_decode(pq):
    pq = np.ascontiguousarray(pq, np.uint8)
    out = np.zeros(65536, np.uint8)
    n = lib().dfko_pq_decode(_p(pq), C.c_uint64(len(pq)), _p(out), C.c_uint64(len(out)))
    if n < 0:
        raise ValueError("bad PQVec stream")
    return out[:n].copy()


def pq_encode(q):
    q = np.ascontiguousarray(q, np.uint8)
    out = np.zeros(2 * len(q) + 16, np.uint8)
    n = lib().dfko_pq_encode(_p(q), C.c_uint32(len(q)), _p(out))
    return out[:n].copy()


def good_len(q, K=48, min_qual=7):
    q = np.ascontiguousarray(q, np.uint8)
    return int(lib().dfko_good_len(_p(q), C.c_uint32(len(q)), C.c_uint(K), C.c_uint(min_qual)))


def kmer_from_codes(codes, K):
    codes = np.ascontiguousarray(codes, np.uint8)
    w = np.zeros(2, np.uint64)
    lib().dfko_kmer_from_codes(_p(codes), C.c_uint(K), _p(w))
    return int(w[0]), int(w[1])


def fnv1a16(w0, w1):
    w = np.array([w0, w1], np.uint64)
    return int(lib().dfko_fnv1a16(_p(w)))


def is_rev(w0, w1, K):
    w = np.array([w0, w1], np.uint64)
    return bool(lib().dfko_is_rev(_p(w), C.c_uint(K)))


def rc(w0, w1, K):
    w = np.array([w0, w1], np.uint64); o = np.zeros(2, np.uint64)
    lib().dfko_rc(_p(w), C.c_uint(K), _p(o))
    return int(o[0]), int(o[1])


def ctx_rc(c):
    return int(lib().dfko_ctx_rc(C.c_uint8(c)))


def spectrum_json(hist):
    hist = np.ascontiguousarray(hist, np.int64)
    need = lib().dfko_spectrum_json(_p(hist), C.c_uint64(len(hist)), None, C.c_uint64(0))
    buf = C.create_string_buffer(int(need) + 1)
    lib().dfko_spectrum_json(_p(hist), C.c_uint64(len(hist)), buf, C.c_uint64(need + 1))
    return buf.raw[:need].decode()
