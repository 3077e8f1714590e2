"""oracle/paths_oracle.py -- TEST INFRASTRUCTURE: CPU restatement of read pathing, for small cases
(pure-Python loops).  Only tests/ may import it.

What it restates (all reference paths relative to lib/assembly/src):
  paths/long/BuildReadQGraph48.cc:685-733     Pather::path: a read as a list of path parts -- runs of k-mers that are not in
                                              the dictionary (gaps) and runs that follow one unipath edge
  paths/long/BuildReadQGraph48.cc:593-682     EdgeLoc / PathPart (offsets of reverse-complement parts, isSameEdge,
                                              isConformingCapturedGap with its unsigned arithmetic)
  paths/long/BuildReadQGraph48.cc:796-803     Pather::isJoinable
  paths/long/BuildReadQGraph48.cc:1212-1317   HBVPather::algorithmTwo -- the aligner StageBuildGraph selects
                                              (10X/runstages/RunStages.cc:389-390: useNewAligner = True)
  paths/long/BuildReadQGraph48.cc:1365-1402   pathPartsToReadPath
  paths/long/ExtendReadPath.cc:15-378         scoreLeftOverlap / scoreRightOverlap (with `unsigned -= double`),
                                              attemptLeftwardExtension / attemptRightwardExtension
  paths/long/ReadPath.h:56-63                 the a.paths element: i32 offset, u32 lastSkip (always 0), i32 edge ids
  10X/WriteFiles.cc:78-82                     a.paths = ReadPathVec::WriteAll (feudal file, header bytes 0/24/4)

  10X/PathsIndex.cc:23-146                    writePathsIndex: a.paths.inv (per edge the reads whose path holds it, ascending; a read
                                              that crosses the edge twice is listed twice) and a.countsb (reads per edge, an edge
                                              and its involution summed)
  10X/SecretOps.cc:410-566                    MarkDups: a.dup

Pinned by tests/golden/*/a.paths (written by oracle/_ref/refdrv graph: the reference's KmerDict::findEntry, KMer,
CF<K>::isRC, bvec iterators, digraphE<basevector> and the ReadPathVec feudal writer; glue restated in
oracle/ref_driver.cc and oracle/ref_graph.cc) in tests/test_paths_oracle.py.
"""
import struct

import numpy as np

from oracle import graph_oracle as go

MAX_JITTER = 3          # HBVPather::MAX_JITTER (:1404)
GAP = None              # edge of a gap part


class Part:
    """PathPart (:622-682): edge = canonical edge index or None for a gap; off = k-mer offset on the edge in the
    orientation the read runs along it; ln = k-mers covered; elen = k-mers on the edge (0 for a gap)."""
    __slots__ = ("edge", "rc", "off", "ln", "elen")

    def __init__(self, edge, rc, off, ln, elen):
        self.edge, self.rc, self.off, self.ln, self.elen = edge, rc, off, ln, elen

    @staticmethod
    def gap(n):
        return Part(GAP, False, 0, n, 0)

    def is_gap(self): return self.elen == 0
    def end_off(self): return self.off + self.ln
    def same_edge(self, o): return self.edge == o.edge and self.rc == o.rc


def path_parts(read, K, place, edges):
    """Pather::path (:685-733).  read: bytes of base codes."""
    n = len(read)
    if n < K:
        return [Part.gap(n)]
    mask = (1 << (2 * K)) - 1
    parts = []

    def find(km):                       # KmerDict::findEntry (kmers/ReadPather.h:222-225)
        r = go._rc(km, K)
        return place.get(r if r < km else km)

    def kmer_at(p):
        v = 0
        for b in read[p:p + K]:
            v = (v << 2) | b
        return v

    at, stop = 0, n - K + 1
    while at != stop:
        km = kmer_at(at)
        hit = find(km)
        if hit is None:
            missed = 1
            nxt = at + K
            at += 1
            while nxt != n:
                km = ((km << 2) | read[nxt]) & mask
                nxt += 1
                hit = find(km)
                if hit is not None:
                    break
                missed += 1
                at += 1
            parts.append(Part.gap(missed))
        if hit is not None:
            e, off = hit
            edge = edges[e]
            # CF<K>::isRC (dna/CanonicalForm.h:84-91): the two k-mers are equal or reverse complements of each other
            rc = bytes(read[at:at + K]) != edge[off:off + K]
            ln = 1
            if not rc:
                i, j = at + K, off + K
                while i < n and j < len(edge) and read[i] == edge[j]:
                    ln += 1; i += 1; j += 1
            else:
                r = go.rc_seq(edge)
                off = len(edge) - off            # first base after the k-mer, on the reverse complement
                i, j = at + K, off
                while i < n and j < len(r) and read[i] == r[j]:
                    ln += 1; i += 1; j += 1
                off -= K
            parts.append(Part(e, rc, off, ln, len(edge) - K + 1))
            at += ln
    return parts


def _u32(x): return x & 0xFFFFFFFF


def _i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


class Pather:
    def __init__(self, graph, K):
        """graph: the dict graph_oracle.run returns (edges, place, hbv)."""
        self.K = K
        self.edges = graph["edges"]
        self.place = graph["place"]
        h = self.h = graph["hbv"]
        self.to_left, self.to_right = h.to_left_right()
        self.rcs = {}

    # ---- pieces of HBVPather / Pather
    def hbv_edge(self, p): return (self.h.rev if p.rc else self.h.fwd)[p.edge]
    def kmers_of(self, e): return len(self.h.edges[e]) - self.K + 1

    def conforming(self, before, gap, after):          # isConformingCapturedGap (:657-666)
        dist = _u32(after.off - before.end_off())
        if not before.same_edge(after):
            dist = _u32(dist + before.elen)
        return _u32(abs(_i32(gap.ln - dist))) <= MAX_JITTER

    def oriented(self, p):
        if not p.rc:
            return self.edges[p.edge]
        if p.edge not in self.rcs:
            self.rcs[p.edge] = go.rc_seq(self.edges[p.edge])
        return self.rcs[p.edge]

    def joinable(self, a, b):                          # isJoinable (:796-803)
        if a.edge == b.edge:
            return True
        klo = self.K - 1
        return self.oriented(a)[-klo:] == self.oriented(b)[:klo]

    def to_read_path(self, parts):                     # pathPartsToReadPath (:1365-1402) -> (offset, [edges])
        path, last = [], None
        for p in parts:
            if p.is_gap() or (last is not None and last.same_edge(p)):
                continue
            path.append(self.hbv_edge(p)); last = p
        if not path:
            return 0, path
        if not parts[0].is_gap():
            return parts[0].off, path
        return parts[1].off - parts[0].ln, path

    @staticmethod
    def score(read, q, start, edge, K, left):          # scoreLeftOverlap / scoreRightOverlap (ExtendReadPath.cc:15-113)
        total = penalty = 0
        n = len(read)
        r = start - 1 if left else n - start
        e = len(edge) - K if left else K - 1
        step = -1 if left else 1
        while 0 <= r < n and 0 <= e < len(edge):
            if read[r] != edge[e]:
                penalty = _u32(penalty + (20 if q[r] == 2 else int(q[r])))
                total = _u32(total + penalty)
            elif penalty > 0:
                penalty = int(float(penalty) - 0.2 * float(penalty))      # `unsigned -= double`: IEEE double, truncated
            r += step; e += step
        while 0 <= r < n:
            total = _u32(total + 10); r += step
        return total

    def extend(self, off, path, read, q, left):        # attemptLeft/RightwardExtension (ExtendReadPath.cc:130-378)
        h, K = self.h, self.K
        if not path:
            return None
        if left:
            if off >= 0:
                return None
            hang = -off
        else:
            hang = len(read) + off - sum(self.kmers_of(e) for e in path) - (K - 1)
        if hang < 10:
            return None
        if left:
            v = self.to_left[path[0]]; cand = h.to_e[v]; far = h.to[v]
            dead = lambda w: len(h.to[w]) == 0 and len(h.frm[w]) == 1
            fan = lambda w: len(h.to[w])
        else:
            v = self.to_right[path[-1]]; cand = h.frm_e[v]; far = h.frm[v]
            dead = lambda w: len(h.frm[w]) == 0 and len(h.to[w]) == 1
            fan = lambda w: len(h.frm[w])
        hanging = [dead(w) for w in far]
        reaches = [len(h.edges[e]) - (K - 1) >= hang for e in cand]
        short_to = [w for w, hg, rch in zip(far, hanging, reaches) if not rch and not hg]
        if len(cand) != 1 and short_to:
            if any(reaches):
                return None
            u = sorted(set(short_to))
            if len(u) != 1 or fan(u[-1]) != 1:
                return None
        best, least = -1, 0xFFFFFFFF
        for e, hg in zip(cand, hanging):
            if not hg or len(cand) == 1:
                s = self.score(read, q, hang, h.edges[e], K, left)
                if s < least:
                    least, best = s, e
        if best == -1 or least > hang * 10:
            return None
        if left:
            return off + self.kmers_of(best), [best] + path
        return off, path + [best]

    def read_path(self, read, q):
        """HBVPather::algorithmTwo (:1212-1317) -> (offset, [HBV edge ids])."""
        parts = path_parts(read, self.K, self.place, self.edges)
        h = self.h
        kept = []
        for p in parts:
            if not p.is_gap():
                e = self.hbv_edge(p)
                vl, vr = self.to_left[e], self.to_right[e]
                if len(h.to[vl]) == 0 and len(h.to[vr]) > 1 and len(h.frm[vr]) > 0 and p.elen <= 100:
                    p = Part.gap(p.ln)
            if p.is_gap() and kept and kept[-1].is_gap():
                kept[-1].ln += p.ln
            else:
                kept.append(p)
        parts = kept
        if len(parts) >= 3:
            seeds = 0 if parts[0].is_gap() else 1
            for i in range(1, len(parts) - 1):
                if not parts[i].is_gap():
                    seeds += 1
                    continue
                if self.conforming(parts[i - 1], parts[i], parts[i + 1]) and self.joinable(parts[i - 1], parts[i + 1]):
                    continue
                if seeds > 1:
                    tail = Part.gap(parts[i - 1].ln + sum(p.ln for p in parts[i:]))
                    parts = parts[:i - 1] + [tail]
                else:
                    parts[i].ln += sum(p.ln for p in parts[i + 1:])
                    parts = parts[:i + 1]
                break
        if parts[-1].is_gap() and len(parts) > 1:
            seed = parts[-2]
            if seed.off == 0 and seed.ln <= 5:
                parts = parts[:-2] + [Part.gap(parts[-1].ln + seed.ln)]
        elif not parts[-1].is_gap():
            seed = parts[-1]
            if seed.off == 0 and seed.ln <= 5:
                parts[-1] = Part.gap(seed.ln)
        off, path = self.to_read_path(parts)
        for i in range(len(path) - 1):
            if self.to_right[path[i]] != self.to_left[path[i + 1]]:
                path = path[:i + 1]
                break
        for left in (True, False):
            while True:
                r = self.extend(off, path, read, q, left)
                if r is None:
                    break
                off, path = r
        return off, path


def unpack_reads(rs):
    """The arrays the ABI takes (packed 2-bit reads + PQVec qualities) -> lists of base-code bytes and quality arrays."""
    from oracle import pyoracle
    reads, quals = [], []
    for i in range(int(rs["n_reads"]) if "n_reads" in rs else len(rs["read_len"])):
        L = int(rs["read_len"][i])
        b = np.asarray(rs["packed"][int(rs["base_off"][i]):int(rs["base_off"][i]) + (L + 3) // 4], np.uint8)
        codes = ((b[:, None] >> np.array([0, 2, 4, 6], np.uint8)) & 3).reshape(-1)[:L].astype(np.uint8)
        reads.append(codes.tobytes())
        quals.append(pyoracle.pq_decode(rs["pq_bytes"][int(rs["pq_off"][i]):int(rs["pq_off"][i + 1])]))
    return reads, quals


def paths_file(paths):
    """a.paths: feudal file of ReadPath (ReadPath.h:56-63; FeudalFileWriter.cc:26-121): 24-byte control block (n, flags 1,
    sizeofFixed 0, sizeofX 24, sizeofA 4), per read {i32 offset, u32 0, i32 edges...}, n+1 absolute u64 offsets."""
    var = b"".join(struct.pack("<iI", off, 0) + np.asarray(p, "<i4").tobytes() for off, p in paths)
    sizes = np.array([8 + 4 * len(p) for _, p in paths], np.uint64)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64) + np.uint64(24)
    n = len(paths)
    var_tab = 24 + len(var)
    head = struct.pack("<IBBBBQQ", n, 1, 0, 24, 4, var_tab, var_tab + 8 * (n + 1))
    return head + var + offs.astype("<u8").tobytes()


def run(reads, quals, graph, K):
    """-> {"paths": [(offset, [edges])...], "file": bytes of a.paths}"""
    p = Pather(graph, K)
    paths = [p.read_path(r, q) for r, q in zip(reads, quals)]
    return dict(paths=paths, file=paths_file(paths))


# ----------------------------------------------------------------------------------------------- row f-4
def paths_index(paths, inv):
    """writePathsIndex (10X/PathsIndex.cc:23-146) -> {"a.paths.inv": bytes, "a.countsb": bytes}.  inv = the graph's
    involution (a.inv).  (The reference walks the edges in 30 chunks and overruns below ~870 edges; this is what it writes
    when it does not.)"""
    n_edges = len(inv)
    pairs = sorted((e, i) for i, (_, p) in enumerate(paths) for e in p)
    lists = [[] for _ in range(n_edges)]
    for e, i in pairs:
        lists[e].append(i)
    var = b"".join(np.asarray(l, "<u8").tobytes() for l in lists)
    offs = np.concatenate([[0], np.cumsum([8 * len(l) for l in lists])]).astype(np.uint64) + np.uint64(24)
    var_tab = 24 + len(var)
    head = struct.pack("<IBBBBQQ", n_edges, 1, 0, 16, 8, var_tab, var_tab + 8 * (n_edges + 1))       # feudal file of SerfVec<unsigned long>
    counts = [len(l) for l in lists]
    for e in range(n_edges):
        if e < inv[e]:
            counts[e] = counts[inv[e]] = counts[e] + counts[inv[e]]
    countsb = b"BINWRITE" + struct.pack("<QQ", 1, n_edges) + np.asarray(counts, "<i4").tobytes()       # vec<vec<int>> with one row
    return {"a.paths.inv": head + var + offs.astype("<u8").tobytes(), "a.countsb": countsb}


def mark_dups(paths, reads, quals):
    """MarkDups (10X/SecretOps.cc:410-566) -> bytes of a.dup (vec<Bool>, one per PAIR).  Reads with the same first edge, the
    same offset on it and the same first five bases of their MATE are one group; the member whose pair has the largest
    quality sum stays (the lowest read id among equals), the pairs of the others are marked."""
    n = len(paths)
    groups = {}
    for i, (off, p) in enumerate(paths):
        if not p:
            continue
        head = 0
        for b in reads[i ^ 1][:5]:
            head = 4 * head + b
        groups.setdefault((p[0], off, head), []).append(i)
    dup = np.zeros(n // 2, np.uint8)
    qsum = lambda i: int(np.asarray(quals[i], np.int64).sum() + np.asarray(quals[i ^ 1], np.int64).sum())
    for ids in groups.values():
        if len(ids) < 2:
            continue
        best = max(ids, key=lambda i: (qsum(i), -i))
        for i in ids:
            if i != best:
                dup[i // 2] = 1
    return b"BINWRITE" + struct.pack("<Q", n // 2) + dup.tobytes()
