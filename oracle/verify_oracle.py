"""oracle/verify_oracle.py -- TEST INFRASTRUCTURE: the CPU forms of dfk_paths_verify and dfk_paths_digest (include/dfk.h),
computed from the FILES of a.<K>/ alone.  Only tests/ may import it.

The library's verifier (csrc/dfk_check_kernels.h, k_path_verify) looks at what stays on the device; this one looks at what
the reference's own writers wrote for the same input (tests/golden/graph_*/: a.fastb, a.to_left, a.to_right, a.paths, ...) and
must arrive at the same eight counters -- which pins the device verifier before it is trusted at sizes no oracle reaches.

What a ReadPath means (paths/long/ReadPath.h:20-63): `offset` = position of the read's first base on the path's first edge
(negative: the read starts in front of it), the edges consecutive in the graph.  So the k-mer at read position p sits at path
coordinate offset + p, counted in k-mers along the concatenated edges (consecutive edges overlap by K-1 bases: an edge of L bases
contributes L-K+1 k-mers).  Every solid k-mer lies on exactly one canonical edge (buildEdges, BuildReadQGraph48.cc:505-530), so
as a string it occurs once among the HBV edges (twice if it is its own reverse complement)."""
import struct

import numpy as np

M64 = (1 << 64) - 1


def mix(x):
    x &= M64
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & M64
    x ^= x >> 27; x = (x * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def _mixv(x):
    x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
    x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def seq_digest(v, salt):
    """k_digest_seq: (sum, xor, plain sum) over i of mix(mix(i + salt) ^ v[i])."""
    v = np.asarray(v).astype(np.uint64)
    if not len(v):
        return 0, 0, 0
    with np.errstate(over="ignore"):
        i = np.arange(len(v), dtype=np.uint64)
        h = _mixv(_mixv(i + np.uint64(salt)) ^ v)
        x = _mixv(h + np.uint64(0xD1B54A32D192ED03))
        return int(h.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(x)), int(v.sum(dtype=np.uint64))


def paths_digest(paths):
    """k_paths_digest over [(offset, [edges])...]: the element as a.paths holds it -- i32 offset, u32 lastSkip = 0, i32 edges."""
    s = x = 0
    for r, (off, edges) in enumerate(paths):
        h = mix(r + 0x9E3779B97F4A7C15)
        for w in [off & 0xFFFFFFFF, 0] + [e & 0xFFFFFFFF for e in edges]:
            h = mix(h ^ w)
        s = (s + h) & M64
        x ^= mix(h + 0xD1B54A32D192ED03)
    return s, x


def decode_feudal_lists(b, dtype):
    n = int.from_bytes(b[:4], "little")
    var_tab = int.from_bytes(b[8:16], "little")
    offs = np.frombuffer(b, "<u8", n + 1, var_tab)
    return [np.frombuffer(b, dtype, (int(offs[i + 1]) - int(offs[i])) // np.dtype(dtype).itemsize, int(offs[i])) for i in range(n)]


def expected_check_words(paths, inv_file, countsb_file, dup_file, inv):
    """What dfk_paths_digest must return for these files (the DFK_CK_* words that depend on files), as a dict."""
    out = {}
    out["PATHS_SUM"], out["PATHS_XOR"] = paths_digest(paths)
    out["N_READS"] = len(paths)
    out["N_PLACED"] = sum(1 for _, p in paths if p)
    out["N_PATH_EDGES"] = sum(len(p) for _, p in paths)
    if inv_file is not None:
        lists = decode_feudal_lists(inv_file, "<u8")
        flat = np.concatenate(lists) if lists else np.zeros(0, np.uint64)
        starts = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.uint64)
        out["INV_SUM"], out["INV_XOR"], _ = seq_digest(flat, 0x1111)
        out["INV_STARTS"] = seq_digest(starts, 0x2222)[0]
        out["INV_ENTRIES"] = int(len(flat))
        csb = np.frombuffer(countsb_file, "<i4", offset=24)
        out["COUNTSB_DIGEST"], _, out["COUNTSB_SUM"] = seq_digest(csb.astype(np.uint32), 0x3333)
        out["SELF_INVERSE"] = int(sum(len(lists[e]) for e in range(len(lists)) if inv[e] == e))
    if dup_file is not None:
        d = np.frombuffer(dup_file, np.uint8, offset=16)
        out["DUP_DIGEST"], _, out["DUP_MARKED"] = seq_digest(d, 0x4444)
    return out


def rc(s):
    return bytes(3 - b for b in reversed(s))


def verify(hbv_edges, to_left, to_right, paths, reads, K):
    """-> the eight counters of dfk_paths_verify: [placed, broken, hits, consistent, no anchor, all consistent, dictionary bad,
    outside].  hbv_edges: the HBV's edges as base-code bytes (a.fastb, both orientations of every unipath edge)."""
    where = {}
    for e, s in enumerate(hbv_edges):
        for j in range(len(s) - K + 1):
            where.setdefault(s[j:j + K], []).append((e, j))
    nk = [len(s) - K + 1 for s in hbv_edges]
    E = len(hbv_edges)
    c = [0] * 8
    for (off, edges), read in zip(paths, reads):
        if not edges:
            continue
        c[0] += 1
        broken = any(e < 0 or e >= E for e in edges) or any(to_right[a] != to_left[b] for a, b in zip(edges, edges[1:]))
        if not broken and off >= nk[edges[0]]:
            broken = True
        if broken:
            c[1] += 1
            continue
        if len(read) < K:
            c[4] += 1
            continue
        total = sum(nk[e] for e in edges)
        hits = good = 0
        for at in range(len(read) - K + 1):
            places = where.get(read[at:at + K])
            if places is None:
                continue
            coord = off + at
            if coord < 0 or coord >= total:
                c[7] += 1
                continue
            hits += 1
            before, j = 0, 0
            while coord >= before + nk[edges[j]]:
                before += nk[edges[j]]; j += 1
            good += (edges[j], coord - before) in places
        c[2] += hits; c[3] += good
        if not good:
            c[4] += 1
        if good == hits:
            c[5] += 1
    return c


def load_graph_dir(d, K):
    """a.fastb, a.to_left, a.to_right, a.inv of a directory -> (edges as base-code bytes, to_left, to_right, inv)"""
    import os
    from superplus_amd import feudal
    packed, base_off, read_len = feudal.read_fastb(os.path.join(d, "a.fastb"))
    edges = []
    for i, L in enumerate(read_len):
        b = np.asarray(packed[int(base_off[i]):int(base_off[i]) + (int(L) + 3) // 4], np.uint8)
        edges.append(((b[:, None] >> np.array([0, 2, 4, 6], np.uint8)) & 3).reshape(-1)[:int(L)].astype(np.uint8).tobytes())
    def vec(name):
        b = open(os.path.join(d, name), "rb").read()
        n = struct.unpack_from("<Q", b, 8)[0]
        return [int(x) for x in np.frombuffer(b, "<i4", n, 16)]
    return edges, vec("a.to_left"), vec("a.to_right"), vec("a.inv")
