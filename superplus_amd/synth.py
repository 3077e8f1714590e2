"""Seeded synthetic stLFR read sets in the reference's in-memory layout (harness only).

SURVEY.md section 8d: random ACGT genome, pairs of 2xL bp with insert U[250,450), random
strand, substitution errors, constant-Q bodies with an optional Q2 3' tail, ~10 % pairs
unbarcoded (barcode 0), the rest grouped into barcodes whose reads come from a few long
"molecules"; reads ordered unbarcoded first then by barcode, R1 even / R2 odd, exactly
the order DF's LoadData produces (10X/DfTools.cc:99-160).

Everything is torch tensor plumbing so the same code fills HBM directly for bench.py
(device="cuda") and runs on the CPU for tests.  No reference code involved.
"""
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class ReadSet:
    packed: torch.Tensor      # u8[sum ceil(L/4)]   .fastb var data
    base_off: torch.Tensor    # i64[n+1]
    read_len: torch.Tensor    # i32[n]
    pq_bytes: torch.Tensor    # u8[]                .qualp var data
    pq_off: torch.Tensor      # i64[n+1]
    bc: torch.Tensor          # i32[n]              0 = unbarcoded
    bci: np.ndarray           # i64[n_barcodes+1]   .bci
    n_reads: int

    def head(self, n_pairs):
        """The first n_pairs pairs as a read set of their own (views of the same tensors): the barcode index cut at the last
        read kept -- a barcode cut in two keeps an even number of reads, pairs being whole."""
        n = 2 * int(n_pairs)
        if n >= self.n_reads:
            return self
        bci = [int(b) for b in self.bci if int(b) <= n]
        if bci[-1] != n:
            bci.append(n)
        return ReadSet(packed=self.packed, base_off=self.base_off[: n + 1], read_len=self.read_len[:n], pq_bytes=self.pq_bytes,
                       pq_off=self.pq_off[: n + 1], bc=self.bc[:n], bci=np.asarray(bci, np.int64), n_reads=n)

    def numpy(self):
        return dict(
            packed=self.packed.cpu().numpy(),
            base_off=self.base_off.cpu().numpy().astype(np.uint64),
            read_len=self.read_len.cpu().numpy().astype(np.uint32),
            pq_bytes=self.pq_bytes.cpu().numpy(),
            pq_off=self.pq_off.cpu().numpy().astype(np.uint64),
            bc=self.bc.cpu().numpy().astype(np.int32),
            bci=self.bci,
            n_reads=self.n_reads,
        )


def make_genome(size, seed, device="cpu", repeat_frac=0.0, family_copies=0, family_len=300, family_div=0.10,
                low_complexity_frac=0.0):
    """family_copies > 0 plants that many diverged copies (substitution rate family_div) of ONE random
    family_len-bp element, Alu-like: a few minimizers then own a large share of the k-mers (hot buckets).
    low_complexity_frac > 0 overwrites that share of the genome with 200-bp microsatellite stretches (units of 1-6 bp:
    poly-A, (AT)n, (CAG)n ...), whose k-mers are few and heavily repeated.  Both are planted on `device`."""
    dev = torch.device(device)
    if dev.type != "cpu":
        g = torch.Generator(device=dev).manual_seed(seed)
        genome = torch.randint(0, 4, (size,), generator=g, dtype=torch.uint8, device=dev)
    else:
        g = torch.Generator(device="cpu").manual_seed(seed)
        genome = torch.randint(0, 4, (size,), generator=g, dtype=torch.uint8)
    if repeat_frac > 0 and size > 4000:
        # plant copies of a few 1-2 kb segments so some k-mers have high multiplicity
        n_rep = max(1, int(size * repeat_frac / 1500))
        src = torch.randint(0, size - 2000, (n_rep,), generator=g, device=dev)
        dst = torch.randint(0, size - 2000, (n_rep,), generator=g, device=dev)
        ln = torch.randint(1000, 2000, (n_rep,), generator=g, device=dev)
        for s, d, l in zip(src.tolist(), dst.tolist(), ln.tolist()):
            genome[d : d + l] = genome[s : s + l].clone()
    if family_copies > 0 and size > 4 * family_len:
        elem = torch.randint(0, 4, (family_len,), generator=g, dtype=torch.uint8, device=dev)
        ar = torch.arange(family_len, device=dev)
        # (copies sit in distinct family_len-wide slots: overlapping scatter writes have no defined winner on a GPU,
        # and the same seed must give the same genome in every process)
        slots = torch.randperm(size // family_len, generator=g, device=dev)[:family_copies] * family_len
        family_copies = int(slots.numel())
        for a in range(0, family_copies, 1 << 18):              # in slabs: a million copies are 300 M cells
            m = min(1 << 18, family_copies - a)
            pos = slots[a : a + m]
            copies = elem[None, :].repeat(m, 1)
            hit = torch.rand(copies.shape, generator=g, device=dev) < family_div
            sub = torch.randint(1, 4, copies.shape, generator=g, dtype=torch.uint8, device=dev)
            copies = torch.where(hit, (copies + sub) & 3, copies)
            genome[(pos[:, None] + ar[None, :]).reshape(-1)] = copies.reshape(-1)
            del pos, copies, hit, sub
    if low_complexity_frac > 0 and size > 4000:
        span = 200
        n = max(1, int(size * low_complexity_frac / span))
        ar = torch.arange(span, device=dev)
        slots = torch.randperm(size // span, generator=g, device=dev)[:n] * span      # distinct slots, as above
        n = int(slots.numel())
        for a in range(0, n, 1 << 18):
            m = min(1 << 18, n - a)
            pos = slots[a : a + m]
            unit_len = torch.randint(1, 7, (m,), generator=g, device=dev)
            unit = torch.randint(0, 4, (m, 6), generator=g, dtype=torch.uint8, device=dev)
            cells = torch.gather(unit, 1, ar[None, :] % unit_len[:, None])
            genome[(pos[:, None] + ar[None, :]).reshape(-1)] = cells.reshape(-1)
            del pos, unit_len, unit, cells
    return genome


def make_reads(genome, n_pairs, seed, read_len=100, err=0.005, unbar_frac=0.10,
               pairs_per_barcode=20, tails=(0, 0, 0, 5, 15), quals=(30, 35, 37),
               chunk=1 << 22, ragged_frac=0.0):
    """-> ReadSet on genome.device.  All random draws happen on that device (a seeded torch.Generator), so a
    30x human-scale set is generated in HBM without touching the host."""
    dev = genome.device
    G = genome.numel()
    L = read_len
    assert G > 600 and L % 4 == 0 and L <= 255       # (one PQVec block per quality run: nQs is a byte)
    g = torch.Generator(device=dev).manual_seed(seed)
    ri = lambda lo, hi, n: torch.randint(lo, hi, (n,), generator=g, device=dev)
    n_unbar = int(n_pairs * unbar_frac)
    n_bar = n_pairs - n_unbar
    n_bc = max(1, n_bar // pairs_per_barcode) if n_bar else 0

    # barcode of every pair, sorted so that reads are grouped by barcode (0 = unbarcoded, first)
    pair_bc = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
    if n_bar:
        pair_bc[n_unbar:] = torch.sort(ri(1, n_bc + 1, n_bar))[0].to(torch.int32)
    counts = torch.bincount(pair_bc, minlength=n_bc + 1) * 2
    bci = np.concatenate([[0], np.cumsum(counts.cpu().numpy())]).astype(np.int64)
    del counts

    # per-barcode molecules: 2 per barcode, 10-50 kb
    if n_bar:
        mol_len = torch.randint(10000, 50000, (n_bc + 1, 2), generator=g, device=dev).clamp(max=max(500, G - 500))
        mol_start = (torch.rand((n_bc + 1, 2), generator=g, device=dev) * (G - mol_len).clamp(min=1)).long()

    n_reads = 2 * n_pairs
    nb = L // 4
    packed = torch.empty((n_reads, nb), dtype=torch.uint8, device=dev)
    ar = torch.arange(L, device=dev)
    for a in range(0, n_pairs, chunk):
        b = min(n_pairs, a + chunk)
        m = b - a
        ins = ri(250, 450, m)
        f = ri(0, G - 450, m)
        st = ri(0, 2, m)
        if n_bar:
            pbc = pair_bc[a:b].long()
            which = ri(0, 2, m)
            ms = mol_start[pbc, which]; ml = mol_len[pbc, which]
            inside = ms + (torch.rand(m, generator=g, device=dev) * (ml - 450).clamp(min=1)).long()
            f = torch.where(pbc > 0, inside.clamp(max=G - 451), f)
            del pbc, which, ms, ml, inside
        fw = genome[(f[:, None] + ar[None, :])]                                  # left read, forward
        rv = 3 - genome[(f + ins - 1)[:, None] - ar[None, :]]                    # right read, reverse-complement
        r1 = torch.where(st[:, None] == 0, fw, rv)
        r2 = torch.where(st[:, None] == 0, rv, fw)
        del fw, rv
        both = torch.stack([r1, r2], dim=1).reshape(-1, L)
        del r1, r2
        hit = torch.rand(both.shape, device=dev, generator=g) < err
        sub = torch.randint(1, 4, both.shape, device=dev, generator=g, dtype=torch.uint8)
        both = torch.where(hit, (both + sub) & 3, both)
        del hit, sub
        c = both.reshape(-1, nb, 4)
        packed[2 * a : 2 * b] = c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)
        del both, c

    # PQVec: [L-tail x Q][tail x Q2] as nBits=0 blocks {nQs, minQ<<3 (low byte), minQ>>5} + the 0 terminator.
    # ragged_frac of the reads get per-base qualities instead: their body is ONE nBits=2 block (values minQ..minQ+3,
    # random; 17 header bits then 2 bits per quality, LSB-first -- feudal/PQVec.cc:87-127), minQ mostly 33, for one
    # such read in sixteen 5: qualities 5..8 straddle MIN_QUAL = 7, so the trim's run rule works base by base.
    tail_choices = torch.tensor(tails, dtype=torch.int64, device=dev)
    qual_choices = torch.tensor(quals, dtype=torch.int64, device=dev)
    tail = tail_choices[ri(0, len(tails), n_reads)]
    qbody = qual_choices[ri(0, len(quals), n_reads)]
    has_tail = tail > 0
    body_bytes = torch.full((n_reads,), 3, dtype=torch.int64, device=dev)
    if ragged_frac > 0:
        ragged = torch.rand(n_reads, generator=g, device=dev) < ragged_frac
        low = ragged & (ri(0, 16, n_reads) == 0)
        qbody = torch.where(ragged, torch.where(low, 5, 33), qbody)
        body_bytes = torch.where(ragged, (2 * (L - tail) + 24) >> 3, body_bytes)
        del low
    pq_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    pq_off[1:] = torch.cumsum(body_bytes + torch.where(has_tail, 3, 0) + 1, 0)
    total = int(pq_off[-1])
    if ragged_frac > 0:                      # payload bits are random; headers, tails and terminators are written over them
        pq = torch.randint(0, 256, (total,), generator=g, dtype=torch.uint8, device=dev)
    else:
        pq = torch.zeros(total, dtype=torch.uint8, device=dev)
    o = pq_off[:-1]
    pq[o] = (L - tail).to(torch.uint8)
    if ragged_frac > 0:
        nbits = torch.where(ragged, 2, 0)
        pq[o + 1] = (((qbody << 3) & 0xFF) | nbits).to(torch.uint8)
        pq[o + 2] = torch.where(ragged, (pq[o + 2] & 0xFE) | (qbody >> 5).to(torch.uint8), (qbody >> 5).to(torch.uint8))
        del ragged, nbits
    else:
        pq[o + 1] = ((qbody << 3) & 0xFF).to(torch.uint8)
        pq[o + 2] = (qbody >> 5).to(torch.uint8)
    ot = (o + body_bytes)[has_tail]
    pq[ot] = tail[has_tail].to(torch.uint8)
    pq[ot + 1] = (2 << 3) & 0xFF
    pq[ot + 2] = 0
    pq[pq_off[1:] - 1] = 0
    del tail, qbody, has_tail, o, ot, body_bytes
    bc = torch.repeat_interleave(pair_bc, 2)
    return ReadSet(
        packed=packed.reshape(-1),
        base_off=torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * nb,
        read_len=torch.full((n_reads,), L, dtype=torch.int32, device=dev),
        pq_bytes=pq, pq_off=pq_off, bc=bc, bci=bci, n_reads=n_reads)


def write_fastq_pair(path1, path2, n_pairs, seed, genome_size=4_600_000, read_len=100, err=0.005, unbar_frac=0.10,
                     pairs_per_barcode=20, threads=8, chunk=100_000):
    """Seeded synthetic stLFR fastq.gz pair at BASELINE configs[0]'s shape (names "@r<i>#<b1>_<b2>_<b3>/<mate>", barcode 0_0_0 =
    unbarcoded: split_barcode_PEXXX_42_unsort_reads.pl's format), built as byte matrices in numpy and compressed in chunks by a
    few threads (concatenated gzip members are one valid .gz).  The input of the ParseBarcodedFastqs legs; harness only."""
    import gzip
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, genome_size, dtype=np.uint8)
    L = read_len
    n_bc = max(1, n_pairs // pairs_per_barcode)
    letters = np.frombuffer(b"ACGT", np.uint8)

    def digits(v, width):                                               # [n] ints -> [n, width] ASCII digits, zero padded
        out = np.empty((len(v), width), np.uint8)
        for k in range(width):
            out[:, width - 1 - k] = 48 + (v // 10 ** k) % 10
        return out

    def make_chunk(a):
        b = min(n_pairs, a + chunk); n = b - a
        r = np.random.default_rng([seed, a])
        pos = r.integers(0, genome_size - 2 * L - 400, n)
        ins = L + r.integers(0, 250, n)
        bc = r.integers(1, n_bc + 1, n)
        bc[r.random(n) < unbar_frac] = 0
        b1, b2, b3 = bc % 1537, (bc // 1537) % 1537, (bc // (1537 * 1537)) % 1537
        idx = np.arange(a, b)
        outs = []
        for mate in (1, 2):
            start = pos if mate == 1 else pos + ins
            codes = genome[start[:, None] + np.arange(L)[None, :]]
            if mate == 2:
                codes = 3 - codes[:, ::-1]
            hit = r.random((n, L)) < err
            codes = np.where(hit, (codes + r.integers(1, 4, (n, L))) & 3, codes).astype(np.uint8)
            seq = letters[codes]
            seq[r.random((n, L)) < 0.0005] = ord("N")
            q = np.repeat(r.choice(np.array([30, 35, 37], np.uint8), n)[:, None], L, axis=1)
            tail = r.choice(np.array([0, 0, 0, 5, 15]), n)
            q[np.arange(L)[None, :] >= (L - tail)[:, None]] = 2
            q[r.random((n, L)) < 0.01] = 20
            name = np.concatenate([np.full((n, 2), ord("@"), np.uint8), digits(idx, 10), np.full((n, 1), ord("#"), np.uint8), digits(b3, 4),
                                   np.full((n, 1), ord("_"), np.uint8), digits(b2, 4), np.full((n, 1), ord("_"), np.uint8), digits(b1, 4),
                                   np.full((n, 1), ord("/"), np.uint8), np.full((n, 1), 48 + mate, np.uint8), np.full((n, 1), 10, np.uint8)], axis=1)
            name[:, 1] = ord("r")
            rec = np.concatenate([name, seq, np.full((n, 1), 10, np.uint8), np.full((n, 1), ord("+"), np.uint8), np.full((n, 1), 10, np.uint8),
                                  (q + 33).astype(np.uint8), np.full((n, 1), 10, np.uint8)], axis=1)
            outs.append(gzip.compress(rec.tobytes(), compresslevel=1))
        return outs

    with ThreadPoolExecutor(threads) as ex, open(path1, "wb") as f1, open(path2, "wb") as f2:
        for o1, o2 in ex.map(make_chunk, range(0, n_pairs, chunk)):
            f1.write(o1); f2.write(o2)
