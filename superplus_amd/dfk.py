"""ctypes binding of libdfk.so (include/dfk.h) -- the host-side mirror used by tests, bench.py
and the multi-GPU driver.  The library is HIP-only: if it cannot be loaded, or no gfx950
device is usable, everything here raises; there is no CPU fallback.

Reference interface mirrored: createDict(work_dir, reads, quals, minQual, minFreq,
ignBcBelow, mem_frac, minBC, bcp) -- lib/assembly/src/paths/long/BuildReadQGraph48.cc:211-215.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DFK_LIB") or os.path.join(_HERE, "libdfk.so")      # DFK_LIB: a developer's timing variant (tools/)
ABI_VERSION = 1
F_KEEP_PRE_ADJ = 1
F_KEEP_INPUTS = 2

ENTRY_DTYPE = np.dtype([("w0", "<u8"), ("w1", "<u8"), ("edge_id", "<u4"), ("count_ctx", "<u4"),
                        ("bc", "<i4"), ("pad", "<u4")])

EXPORTS = [
    "dfk_create", "dfk_destroy", "dfk_last_error", "dfk_abi_version", "dfk_count", "dfk_count_bci", "dfk_count_device",
    "dfk_hint_file_range", "dfk_qual_hist", "dfk_paths_sink", "dfk_good_lens", "dfk_spectrum", "dfk_spectrum_json", "dfk_solid_count", "dfk_solid_fetch", "dfk_solid_fetch_unsorted", "dfk_solid_digest",
    "dfk_write_kvec", "dfk_write_kvec_part", "dfk_get_stats", "dfk_shard_begin", "dfk_shard_begin_host", "dfk_shard_plan", "dfk_shard_partition", "dfk_shard_partition_begin", "dfk_shard_partition_end", "dfk_shard_recv_buffer", "dfk_shard_count", "dfk_shard_adj_queries",
    "dfk_shard_adj_answer", "dfk_shard_adj_apply", "dfk_graph_build", "dfk_graph_stats", "dfk_graph_write",
    "dfk_shard_dict_share", "dfk_shard_dict_adopt", "dfk_shard_dict_whole",
    "dfk_paths_build", "dfk_paths_build_device", "dfk_paths_stats", "dfk_paths_write", "dfk_paths_fetch", "dfk_paths_index_write", "dfk_dups_write", "dfk_paths_index_dups_write",
    "dfk_paths_digest", "dfk_paths_verify", "dfk_paths_verify_device", "dfk_pbf_run", "dfk_pbf_result", "dfk_pbf_free",
    "dfk_paths_var_bytes", "dfk_paths_write_part", "dfk_shard_pidx_pairs", "dfk_shard_pidx_write", "dfk_shard_dup_keys", "dfk_shard_dup_answer", "dfk_shard_dup_write",
]

# dfk_paths_digest's words (include/dfk.h, DFK_CK_*) and dfk_paths_verify's counters
CHECK_WORDS = 20
CK = ["PATHS_SUM", "PATHS_XOR", "N_READS", "N_PLACED", "N_PATH_EDGES", "INV_SUM", "INV_XOR", "INV_STARTS", "INV_ENTRIES",
      "COUNTSB_DIGEST", "COUNTSB_SUM", "SELF_INVERSE", "DUP_DIGEST", "DUP_MARKED", "EDGE_KMERS", "N_SOLID", "INV_VIOLATIONS",
      "N_EDGES", "VALID"]
VERIFY = ["placed", "broken", "hits", "consistent", "no_anchor", "all_consistent", "dict_bad", "outside"]


class DfkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dfk error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("K", C.c_uint32), ("min_qual", C.c_uint32), ("min_freq", C.c_uint32),
                ("min_bc", C.c_uint32), ("device", C.c_int32), ("ign_bc_below", C.c_int64),
                ("hbm_budget_bytes", C.c_uint64), ("minimizer_len", C.c_uint32), ("flags", C.c_uint32),
                ("inst_per_item", C.c_uint64), ("reserved", C.c_uint64 * 4)]


class Stats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_inst", C.c_uint64), ("n_records", C.c_uint64), ("n_buckets", C.c_uint64),
                ("n_items", C.c_uint64), ("n_overflow_items", C.c_uint64), ("n_distinct", C.c_uint64),
                ("n_solid", C.c_uint64), ("adj_probes", C.c_uint64),
                ("ms_upload", C.c_float), ("ms_trim", C.c_float), ("ms_part_count", C.c_float),
                ("ms_part_scatter", C.c_float), ("ms_count", C.c_float), ("ms_fallback", C.c_float),
                ("ms_adjacency", C.c_float), ("ms_total", C.c_float),
                ("hbm_bytes_peak", C.c_uint64), ("reserved", C.c_uint64 * 8)]

    def asdict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}
        d["n_passes"] = int(self.reserved[0])
        d["us_graph_device"], d["us_graph_host"], d["us_paths"] = int(self.reserved[1]), int(self.reserved[2]), int(self.reserved[3])
        d["gate_timeouts"] = int(self.reserved[4])
        d["hbm_held"] = int(self.reserved[5])
        d["n_scan_launches"] = int(self.reserved[6])
        return d


_lib = None


def build():
    """Compile libdfk.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DfkError(-2, f"{LIB_PATH} is missing: run `make -C superplus_amd/csrc` (hipcc, gfx950). "
                               "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.dfk_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise DfkError(rc, lib().dfk_last_error().decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Dfk:
    """One context on one GPU.  count(...) takes numpy host arrays; count_device(...) takes
    torch tensors already on this context's device."""

    def __init__(self, K=48, min_qual=7, min_freq=3, min_bc=2, ign_bc_below=0, device=0, hbm_budget_bytes=0,
                 minimizer_len=0, keep_pre_adjacency=False, inst_per_item=0, passes=0, keep_inputs=False):
        cfg = Config(abi_version=ABI_VERSION, K=K, min_qual=min_qual, min_freq=min_freq, min_bc=min_bc, device=device,
                     ign_bc_below=ign_bc_below, hbm_budget_bytes=hbm_budget_bytes, minimizer_len=minimizer_len,
                     flags=(F_KEEP_PRE_ADJ if keep_pre_adjacency else 0) | (F_KEEP_INPUTS if keep_inputs else 0), inst_per_item=inst_per_item)
        cfg.reserved[0] = passes          # forced number of bucket-range passes; 0 = sized from the free HBM
        self._ctx = C.c_void_p()
        _check(lib().dfk_create(C.byref(cfg), C.byref(self._ctx)))
        self.K = K

    def close(self):
        if self._ctx:
            lib().dfk_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count(self, packed, base_off, read_len, pq_bytes, pq_off, bc):
        packed = np.ascontiguousarray(packed, np.uint8); base_off = None if base_off is None else np.ascontiguousarray(base_off, np.uint64)
        read_len = np.ascontiguousarray(read_len, np.uint32); pq_bytes = np.ascontiguousarray(pq_bytes, np.uint8)
        pq_off = np.ascontiguousarray(pq_off, np.uint64)
        bc = None if bc is None else np.ascontiguousarray(bc, np.int32)
        _check(lib().dfk_count(self._ctx, _p(packed), _p(base_off), _p(read_len), _p(pq_bytes), _p(pq_off), _p(bc),
                               C.c_uint64(len(read_len))))

    def count_bci(self, packed, base_off, read_len, pq_bytes, pq_off, bci):
        """dfk_count_bci: the barcode index (as read from .bci) instead of the expanded vector; base_off None = dense bases."""
        packed = np.ascontiguousarray(packed, np.uint8); read_len = np.ascontiguousarray(read_len, np.uint32)
        base_off = None if base_off is None else np.ascontiguousarray(base_off, np.uint64)
        pq_bytes = np.ascontiguousarray(pq_bytes, np.uint8); pq_off = np.ascontiguousarray(pq_off, np.uint64)
        bci = np.ascontiguousarray(bci, np.int64)
        _check(lib().dfk_count_bci(self._ctx, _p(packed), _p(base_off), _p(read_len), _p(pq_bytes), _p(pq_off), _p(bci),
                                   C.c_uint64(len(bci)), C.c_uint64(len(read_len))))

    def count_device(self, packed, base_off, read_len, pq_bytes, pq_off, bc):
        """torch tensors on the GPU: packed u8, base_off i64[n+1], read_len i32[n], pq_bytes u8,
        pq_off i64[n+1], bc i32[n] or None."""
        def dp(t):
            return C.c_void_p(0 if t is None else t.data_ptr())
        for t in (packed, base_off, read_len, pq_bytes, pq_off):
            assert t.is_cuda and t.is_contiguous()
        _check(lib().dfk_count_device(self._ctx, dp(packed), C.c_uint64(packed.numel()), dp(base_off), dp(read_len),
                                      dp(pq_bytes), C.c_uint64(pq_bytes.numel()), dp(pq_off), dp(bc),
                                      C.c_uint64(read_len.numel())))

    def hint_file_range(self, array, fd, file_off=0):
        """dfk_hint_file_range: `array` (a numpy array over a mapped file, e.g. np.memmap) is a mapping of a file.  fd >= 0: its
        bytes are those of descriptor `fd` from `file_off` on, and the host-buffer calls read the file (the mapping may go away
        while they run); fd = -1: they read the memory and drop the pages they have read from the page table.  None forgets."""
        if array is None:
            _check(lib().dfk_hint_file_range(self._ctx, None, C.c_uint64(0), C.c_int(-1), C.c_uint64(0)))
        else:
            _check(lib().dfk_hint_file_range(self._ctx, C.c_void_p(array.ctypes.data), C.c_uint64(array.nbytes), C.c_int(fd), C.c_uint64(file_off)))

    def qual_hist(self, max_len):
        """dfk_qual_hist: int64 [2][max_len][256] from the kept reads."""
        out = np.zeros((2, max_len, 256), np.int64)
        _check(lib().dfk_qual_hist(self._ctx, C.c_uint32(max_len), _p(out)))
        return out

    def good_lens(self):
        n = self.stats()["n_reads"]
        out = np.zeros(n, np.uint32)
        _check(lib().dfk_good_lens(self._ctx, _p(out), C.c_uint64(n)))
        return out

    def spectrum(self):
        h = C.POINTER(C.c_int64)(); n = C.c_uint64()
        _check(lib().dfk_spectrum(self._ctx, C.byref(h), C.byref(n)))
        return np.ctypeslib.as_array(h, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int64)

    def spectrum_json(self):
        need = C.c_uint64()
        _check(lib().dfk_spectrum_json(self._ctx, None, C.c_uint64(0), C.byref(need)))
        buf = C.create_string_buffer(need.value + 1)
        _check(lib().dfk_spectrum_json(self._ctx, buf, C.c_uint64(need.value + 1), C.byref(need)))
        return buf.raw[: need.value].decode()

    def solid_count(self):
        n = C.c_uint64()
        _check(lib().dfk_solid_count(self._ctx, C.byref(n)))
        return n.value

    def solid(self, pre_adjacency=False):
        n = self.solid_count()
        out = np.zeros(n, dtype=ENTRY_DTYPE)
        _check(lib().dfk_solid_fetch(self._ctx, _p(out), C.c_uint64(n), C.c_int(1 if pre_adjacency else 0)))
        return out

    def solid_unsorted(self, pre_adjacency=False):
        n = self.solid_count()
        out = np.zeros(n, dtype=ENTRY_DTYPE)
        _check(lib().dfk_solid_fetch_unsorted(self._ctx, _p(out), C.c_uint64(n), C.c_int(1 if pre_adjacency else 0)))
        return out

    def digest(self, pre_adjacency=False):
        """(sum, xor) digest of the dictionary entries, order-independent (dfk_solid_digest)."""
        out = (C.c_uint64 * 2)()
        _check(lib().dfk_solid_digest(self._ctx, C.c_int(1 if pre_adjacency else 0), out))
        return int(out[0]), int(out[1])

    def write_kvec(self, path, pre_adjacency=False, in_order=False):
        """kmers.kvec image; device order unless in_order (ascending k-mers, sorted on the host)."""
        _check(lib().dfk_write_kvec(self._ctx, path.encode(), C.c_int((1 if pre_adjacency else 0) | (2 if in_order else 0))))

    def graph_build(self):
        """Unipath edges (device) + canonical HyperBasevector (host) of the last count's dictionary."""
        _check(lib().dfk_graph_build(self._ctx))
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().dfk_graph_stats(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
        return dict(n_canonical_edges=a.value, n_vertices=b.value, n_edges=c.value)

    def graph_write(self, directory):
        """a.k, a.hbv, a.hbx, a.to_left, a.to_right, a.inv, a.fastb, a.kmers into `directory` (which must exist)."""
        _check(lib().dfk_graph_write(self._ctx, directory.encode()))

    def paths_build(self, packed=None, base_off=None, read_len=None, pq_bytes=None, pq_off=None):
        """Path every read onto the graph (dfk_graph_build first).  numpy host arrays as for count(); none at all = the reads
        the last count() uploaded and kept (keep_inputs=True)."""
        if packed is None:
            _check(lib().dfk_paths_build(self._ctx, None, None, None, None, None, C.c_uint64(0)))
        else:
            packed = np.ascontiguousarray(packed, np.uint8); base_off = np.ascontiguousarray(base_off, np.uint64)
            read_len = np.ascontiguousarray(read_len, np.uint32); pq_bytes = np.ascontiguousarray(pq_bytes, np.uint8)
            pq_off = np.ascontiguousarray(pq_off, np.uint64)
            _check(lib().dfk_paths_build(self._ctx, _p(packed), _p(base_off), _p(read_len), _p(pq_bytes), _p(pq_off), C.c_uint64(len(read_len))))
        return self.paths_stats()

    def paths_build_device(self, packed, base_off, read_len, pq_bytes, pq_off):
        """torch tensors on the GPU, as for count_device()."""
        def dp(t):
            return C.c_void_p(t.data_ptr())
        _check(lib().dfk_paths_build_device(self._ctx, dp(packed), C.c_uint64(packed.numel()), dp(base_off), dp(read_len),
                                            dp(pq_bytes), C.c_uint64(pq_bytes.numel()), dp(pq_off), C.c_uint64(read_len.numel())))
        return self.paths_stats()

    def paths_stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().dfk_paths_stats(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
        return dict(n_reads=a.value, n_placed=b.value, n_path_edges=c.value)

    def paths_sink(self, path):
        """dfk_paths_sink: the next paths_build streams a.paths' variable data into `path`; paths_write(path) completes it."""
        _check(lib().dfk_paths_sink(self._ctx, None if path is None else path.encode()))

    def paths_write(self, path):
        """a.paths (feudal file of ReadPath) as WriteAssemblyFiles writes it."""
        _check(lib().dfk_paths_write(self._ctx, path.encode()))

    def paths_index_write(self, directory):
        """a.paths.inv and a.countsb (writePathsIndex) into `directory`; None = everything but the files (paths_digest)."""
        _check(lib().dfk_paths_index_write(self._ctx, None if directory is None else directory.encode()))

    def dups_write(self, path):
        """a.dup (MarkDups); returns the number of pairs marked.  None = no file."""
        n = C.c_uint64()
        _check(lib().dfk_dups_write(self._ctx, None if path is None else path.encode(), C.byref(n)))
        return n.value

    def paths_index_dups_write(self, directory, path):
        """dfk_paths_index_dups_write: both steps, a.paths.inv's lists written under the duplicate marking; returns the pairs marked."""
        n = C.c_uint64()
        _check(lib().dfk_paths_index_dups_write(self._ctx, None if directory is None else directory.encode(), None if path is None else path.encode(), C.byref(n)))
        return n.value

    def paths_digest(self):
        """dfk_paths_digest: {name: word} -- content digests of a.paths / a.paths.inv / a.countsb / a.dup and the graph's identities."""
        out = (C.c_uint64 * CHECK_WORDS)()
        _check(lib().dfk_paths_digest(self._ctx, out))
        return {k: int(out[i]) for i, k in enumerate(CK)}

    def paths_verify(self, packed, base_off, read_len):
        """dfk_paths_verify on numpy host arrays (the reads that were pathed): {counter: value}."""
        packed = np.ascontiguousarray(packed, np.uint8); base_off = np.ascontiguousarray(base_off, np.uint64)
        read_len = np.ascontiguousarray(read_len, np.uint32)
        out = (C.c_uint64 * 8)()
        _check(lib().dfk_paths_verify(self._ctx, _p(packed), _p(base_off), _p(read_len), C.c_uint64(len(read_len)), out))
        return {k: int(out[i]) for i, k in enumerate(VERIFY)}

    def paths_verify_device(self, packed, base_off, read_len):
        out = (C.c_uint64 * 8)()
        _check(lib().dfk_paths_verify_device(self._ctx, C.c_void_p(packed.data_ptr()), C.c_uint64(packed.numel()), C.c_void_p(base_off.data_ptr()),
                                             C.c_void_p(read_len.data_ptr()), C.c_uint64(read_len.numel()), out))
        return {k: int(out[i]) for i, k in enumerate(VERIFY)}

    def paths(self):
        """-> (offsets i32[n], first_edge u64[n+1], edges i32[...])"""
        st = self.paths_stats()
        off = np.zeros(st["n_reads"], np.int32); first = np.zeros(st["n_reads"] + 1, np.uint64)
        edges = np.zeros(max(1, st["n_path_edges"]), np.int32)
        _check(lib().dfk_paths_fetch(self._ctx, _p(off), _p(first), _p(edges), C.c_uint64(len(edges))))
        return off, first, edges[: st["n_path_edges"]]

    def stats(self):
        s = Stats()
        _check(lib().dfk_get_stats(self._ctx, C.byref(s)))
        return s.asdict()


def _mix(x):
    x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
    x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def digest_of(entries):
    """numpy form of k_digest (dfk_solid_digest) over 32-byte dictionary entries: (sum, xor).  What a CPU-side answer
    is turned into to be compared with a dictionary that stays on the device."""
    w = np.ascontiguousarray(entries).view(np.uint64).reshape(-1, 4)
    with np.errstate(over="ignore"):
        h = _mix(w[:, 0] ^ _mix(w[:, 1] ^ _mix(w[:, 2] ^ _mix(w[:, 3] + np.uint64(0x9E3779B97F4A7C15)))))
        x = _mix(h + np.uint64(0xD1B54A32D192ED03))
        return int(h.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(x)) if len(x) else 0
