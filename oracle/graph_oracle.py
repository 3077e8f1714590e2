"""oracle/graph_oracle.py -- TEST INFRASTRUCTURE: CPU restatement of the graph half of buildReadQGraph48, for small cases
(pure-Python loops).  Only tests/ may import it.

What it restates (all reference paths relative to lib/assembly/src):
  paths/long/BuildReadQGraph48.cc:320-530   EdgeBuilder / buildEdges: the canonical unipath edges over the solid k-mers
                                            and their contexts AFTER recomputeAdjacencies
  paths/long/HBVFromEdges.cc:106-296        buildHBVFromEdges: vertices = (K-1)-mers at the edge ends, canonical edge
                                            order (length descending, then lexical), queue-ordered numbering of
                                            vertices and edges, both orientations of every edge
  paths/HyperBasevector.cc:121-137,668-680  a.hbv / a.hbx serialisation, Involution
  graph/DigraphTemplate.h:2058-2068         digraphE::AddEdge (adjacency lists kept sorted by neighbour, ties in insertion order)
  10X/WriteFiles.cc:69-101                  the files of a.<K>/

Pinned by tests/golden/graph_* (written by oracle/_ref/refdrv graph: the reference's KmerDict, KMer, digraphE<basevector>,
vecbvec::WriteAll and BinaryWriter; see oracle/ref_graph.cc) in tests/test_graph_oracle.py.
"""
import struct
from collections import deque

import numpy as np


# ----------------------------------------------------------------------------------------------- k-mers as integers
def _kmer_ints(solid, K):
    """(w0, w1) left-aligned words -> 2K-bit integers, base 0 most significant."""
    return [(int(a) << 64 | int(b)) >> (128 - 2 * K) for a, b in zip(solid["w0"], solid["w1"])]


def _rc(x, K):
    m = (1 << (2 * K)) - 1
    x = ~x & m
    r = 0
    for _ in range(K):
        r = (r << 2) | (x & 3)
        x >>= 2
    return r


_CTX_RC = [int("{:08b}".format(i)[::-1], 2) for i in range(256)]          # KMerContext::rc = bit reversal of the byte


def _bases(x, K):
    return bytes((x >> (2 * (K - 1 - i))) & 3 for i in range(K))


def rc_seq(s):
    return bytes(3 - b for b in reversed(s))


def canonical_form(s):
    """getCanonicalForm (dna/CanonicalForm.h:32-46): a sequence of ODD length is REV iff its middle base is G or T
    (it cannot be its own reverse complement); one of even length is compared outside-in with the complement of its
    mirror base, which is the lexical comparison of the sequence with its reverse complement."""
    n = len(s)
    if n & 1:
        return "REV" if s[n // 2] & 2 else "FWD"
    r = rc_seq(s)
    return "PAL" if r == s else ("REV" if r < s else "FWD")


# ----------------------------------------------------------------------------------------------- unipath edges
def build_edges(solid, K):
    """solid: structured array with w0, w1, count_ctx (contexts after recomputeAdjacencies).  -> list of canonical
    edge sequences (bytes of base codes), in no particular order, and {canonical k-mer int: (edge index, offset)}."""
    mask = (1 << (2 * K)) - 1
    kms = _kmer_ints(solid, K)
    ctx_of = {k: int(c) >> 24 for k, c in zip(kms, solid["count_ctx"])}
    place = {}
    edges = []

    def look(k):                                     # -> (canonical k-mer, context as seen in k's orientation)
        r = _rc(k, K)
        if r < k:
            return r, _CTX_RC[ctx_of[r]]
        return k, ctx_of[k]

    def succ(k, b): return ((k << 2) | b) & mask
    def pred(k, b): return (k >> 2) | (b << (2 * K - 2))
    def single(bits): return {1: 0, 2: 1, 4: 2, 8: 3}[bits]
    def n_succ(c): return bin(c & 15).count("1")
    def n_pred(c): return bin(c >> 4).count("1")
    def is_pal(k): return _rc(k, K) == k

    def up_ok(k, c):
        if n_pred(c) != 1:
            return False
        p = pred(k, single(c >> 4))
        return not is_pal(p) and n_succ(look(p)[1]) == 1

    def down_ok(k, c):
        if n_succ(c) != 1:
            return False
        n = succ(k, single(c & 15))
        return not is_pal(n) and n_pred(look(n)[1]) == 1

    def add(seq, on):
        if canonical_form(seq) == "REV":
            seq = rc_seq(seq); on = on[::-1]
        for off, k in enumerate(on):
            assert k not in place, "k-mer already on an edge"
            place[k] = (len(edges), off)
        edges.append(seq)

    def walk(k, c, seq, on):
        nxt = k
        while n_succ(c) == 1:
            b = single(c & 15)
            nxt = succ(nxt, b)
            if is_pal(nxt):
                break
            ck, c = look(nxt)
            if n_pred(c) != 1:
                break
            seq.append(b); on.append(ck)
        s = bytes(seq)
        form = canonical_form(s)
        if form == "PAL":
            assert len(s) == K
        if form != "REV":                            # its mirror image is built from the other end
            add(s, on)

    for k in kms:
        if k in place:
            continue
        c = ctx_of[k]
        if is_pal(k):
            add(_bases(k, K), [k]); continue
        up, down = up_ok(k, c), down_ok(k, c)
        if up and down:
            continue                                 # interior k-mer: its edge is found from an end
        if up:
            r = _rc(k, K)
            walk(r, _CTX_RC[c], bytearray(_bases(r, K)), [k])
        elif down:
            walk(k, c, bytearray(_bases(k, K)), [k])
        else:
            add(_bases(k, K), [k])
    # what is left lies on cycles without branches (simpleCircle, canonicalizeCircle)
    for k in kms:
        if k in place:
            continue
        seq = bytearray(_bases(k, K)); on = [k]
        c = ctx_of[k]; cur = k
        while True:
            b = single(c & 15)
            cur = succ(cur, b)
            ck, c = look(cur)
            if ck == k:
                break
            assert ck not in place
            seq.append(b); on.append(ck)
        idx = min(range(len(on)), key=lambda i: on[i])
        s = bytes(seq)
        if canonical_form(s[idx:idx + K]) == "REV":
            s = rc_seq(s); on = on[::-1]; idx = len(s) - idx - K
        if idx:
            s = s[idx:] + s[K - 1:K + idx - 1]
            on = on[idx:] + on[:idx]
        add(s, on)
    return edges, place


# ----------------------------------------------------------------------------------------------- HBV
class Hbv:
    """The HyperBasevector's content: K, per-vertex sorted adjacency (from / from_edge_obj / to / to_edge_obj), edge
    objects (bytes of base codes), and the canonical-edge -> HBV-edge translation tables."""
    def __init__(self, K):
        self.K = K; self.frm = []; self.frm_e = []; self.to = []; self.to_e = []; self.edges = []
        self.fwd = []; self.rev = []

    def add_vertices(self, n):
        self.frm = [[] for _ in range(n)]; self.frm_e = [[] for _ in range(n)]
        self.to = [[] for _ in range(n)]; self.to_e = [[] for _ in range(n)]

    def add_edge(self, v, w, seq):                   # digraphE::AddEdge: upper_bound insert
        import bisect
        n = len(self.edges)
        self.edges.append(seq)
        i = bisect.bisect_right(self.frm[v], w); self.frm[v].insert(i, w); self.frm_e[v].insert(i, n)
        j = bisect.bisect_right(self.to[w], v); self.to[w].insert(j, v); self.to_e[w].insert(j, n)
        return n

    def to_left_right(self):
        L = [0] * len(self.edges); R = [0] * len(self.edges)
        for v in range(len(self.frm)):
            for w, e in zip(self.frm[v], self.frm_e[v]):
                L[e] = v; R[e] = w
        return L, R

    def involution(self):
        E = len(self.edges)
        x1 = sorted(range(E), key=lambda e: self.edges[e])
        rcs = [rc_seq(s) for s in self.edges]
        x2 = sorted(range(E), key=lambda e: rcs[e])
        inv = [0] * E
        for a, b in zip(x1, x2):
            inv[a] = b
        return inv


def build_hbv(edges, K):
    klo = K - 1
    h = Hbv(K)
    nE = len(edges)
    h.fwd = [-1] * nE; h.rev = [-1] * nE
    if not nE:
        return h
    rcs = [rc_seq(s) for s in edges]
    pal = [r == s for r, s in zip(rcs, edges)]
    order = sorted(range(nE), key=lambda i: (-len(edges[i]), edges[i]))

    def end(i, rc, distal):
        s = rcs[i] if rc else edges[i]
        return s[len(s) - klo:] if distal else s[:klo]

    verts = {}                                       # (K-1)-mer -> [id, incident (edge, rc) in EEComp order]
    for i in order:
        for rc in ((0,) if pal[i] else (0, 1)):
            for distal in (0, 1):
                verts.setdefault(end(i, rc, distal), [-1, []])[1].append((i, rc))
    h.add_vertices(len(verts))
    nxt = 0
    q = deque()
    done = lambda x: (h.rev if x[1] else h.fwd)[x[0]] != -1
    for rc_round in (0, 1):
        for i in order:
            if done((i, rc_round)):
                continue
            q.append((i, rc_round))
            while q:
                x = q.popleft()
                if done(x):
                    continue
                i2, rc = x
                a = verts[end(i2, rc, 0)]
                if a[0] < 0:
                    a[0] = nxt; nxt += 1
                b = verts[end(i2, rc, 1)]
                if b[0] < 0:
                    b[0] = nxt; nxt += 1
                n = h.add_edge(a[0], b[0], rcs[i2] if rc else edges[i2])
                if not rc or pal[i2]:
                    h.fwd[i2] = n
                if rc or pal[i2]:
                    h.rev[i2] = n
                q.extend(y for y in a[1] if not done(y))
                q.extend(y for y in b[1] if not done(y))
    assert nxt == len(verts)
    return h


# ----------------------------------------------------------------------------------------------- files of a.<K>/
def _vec_i32(v):
    return struct.pack("<Q", len(v)) + np.asarray(v, "<i4").tobytes()


def _vecvec_i32(vv):
    return struct.pack("<Q", len(vv)) + b"".join(_vec_i32(v) for v in vv)


def _pack(seq):
    a = np.frombuffer(seq, np.uint8)
    pad = (-len(a)) % 4
    a = np.concatenate([a, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    return (a[:, 0] | (a[:, 1] << 2) | (a[:, 2] << 4) | (a[:, 3] << 6)).astype(np.uint8).tobytes()


def _vec_bvec(edges):                                # vec<basevector>: u64 n, then per element u32 size + packed bytes (FieldVec::writeBinary)
    return struct.pack("<Q", len(edges)) + b"".join(struct.pack("<I", len(s)) + _pack(s) for s in edges)


def fastb_bytes(edges):
    from superplus_amd import feudal
    var = b"".join(_pack(s) for s in edges)
    off = np.concatenate([[0], np.cumsum([(len(s) + 3) // 4 for s in edges])]).astype(np.uint64)
    n = len(edges)
    var_tab = 24 + len(var)
    fixed_off = var_tab + 8 * (n + 1)
    return (feudal.header(n, 4, 16, 1, var_tab, fixed_off) + var + (off + np.uint64(24)).tobytes() +
            np.asarray([len(s) for s in edges], "<u4").tobytes())


def graph_files(h):
    """-> {file name: bytes} for a.<K>/ (WriteFiles.cc:69-101), without the paths and alignment files."""
    L, R = h.to_left_right()
    E = len(h.edges)
    hbv = b"BINWRITE" + struct.pack("<i", h.K) + _vecvec_i32(h.frm) + _vecvec_i32(h.frm_e) + _vecvec_i32(h.to_e) + _vec_bvec(h.edges)
    # HyperBasevectorX = K | digraphEX<basevector> (HyperBasevector.cc:133-137, DigraphTemplate.h:2593-2599,2767-2795):
    # from / to / from_edge_obj / to_edge_obj as MasterVec<SerfVec<int>> (u64 n, then per vertex u32 size + data),
    # the edges, then to_left / to_right
    def serf(vv):
        return struct.pack("<Q", len(vv)) + b"".join(struct.pack("<I", len(v)) + np.asarray(v, "<i4").tobytes() for v in vv)
    hbx = (b"BINWRITE" + struct.pack("<i", h.K) + serf(h.frm) + serf(h.to) + serf(h.frm_e) + serf(h.to_e) + _vec_bvec(h.edges) +
           _vec_i32(L) + _vec_i32(R))
    out = {
        "a.hbx": hbx,
        "a.k": ("%d\n" % h.K).encode(),
        "a.hbv": hbv,
        "a.to_left": b"BINWRITE" + _vec_i32(L),
        "a.to_right": b"BINWRITE" + _vec_i32(R),
        "a.inv": b"BINWRITE" + _vec_i32(h.involution()),
        "a.kmers": b"BINWRITE" + _vec_i32([len(s) - h.K + 1 for s in h.edges]),
        "a.fastb": fastb_bytes(h.edges),
        "fwd_xlat": b"BINWRITE" + _vec_i32(h.fwd),
        "rev_xlat": b"BINWRITE" + _vec_i32(h.rev),
    }
    return out, E


def run(solid, K):
    edges, place = build_edges(solid, K)
    h = build_hbv(edges, K)
    files, _ = graph_files(h)
    return dict(edges=edges, place=place, hbv=h, files=files)
