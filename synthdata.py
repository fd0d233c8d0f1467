"""Deterministic synthetic genomes and long reads (test + bench infrastructure).

SURVEY.md 8(d): no E. coli / GRCh38 FASTA and no ONT FASTQ exist in this image or on
the GPU box, so every config runs on seeded synthetic data of the stated shape:
  * genome: i.i.d. bases at a given GC fraction plus injected repeat families
    (diverged copies of a block) so that multi-occurrence minimizers exist;
  * reads: log-normal lengths scaled to a target N50, uniform start, 50/50 strand,
    i.i.d. substitution / insertion / deletion errors (ONT-like 6 % by default).
Everything is numpy on PCG64 streams; the same seed gives the same bytes.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def random_codes(rng, n, gc=0.5):
    """n bases as codes 0..3 (A,C,G,T) with P(G or C) = gc."""
    r = rng.random(n)
    at = rng.integers(0, 2, n, dtype=np.uint8)           # picks A/T or C/G
    is_gc = r < gc
    return np.where(is_gc, 1 + at, 3 * at).astype(np.uint8)  # gc: C(1)/G(2); at: A(0)/T(3)


def mutate(codes, rng, sub, ins, dele):
    """apply i.i.d. errors to a code array; returns the mutated code array"""
    n = len(codes)
    if n == 0:
        return codes.copy()
    r = rng.random(n)
    keep = r >= dele
    is_sub = keep & (r < dele + sub)
    out = codes.copy()
    ns = int(is_sub.sum())
    if ns:
        out[is_sub] = (out[is_sub] + rng.integers(1, 4, ns, dtype=np.uint8)) & 3
    n_ins = (rng.random(n) < ins).astype(np.int64)
    counts = keep.astype(np.int64) + n_ins
    idx = np.repeat(np.arange(n), counts)
    res = out[idx]
    ends = np.cumsum(counts)
    ins_pos = ends[n_ins == 1] - 1
    if len(ins_pos):
        res[ins_pos] = rng.integers(0, 4, len(ins_pos), dtype=np.uint8)
    return res


def make_genome(seed, contig_lens, gc=0.508, repeats=((5000, 7, 0.01), (1300, 20, 0.01)), n_runs=0):
    """Returns list of uint8 code arrays (0..3, 4 = N).  `repeats` = (block_len, copies, divergence)."""
    rng = _rng(seed)
    contigs = [random_codes(rng, int(n), gc) for n in contig_lens]
    total = sum(len(c) for c in contigs)
    for blen, copies, div in repeats:
        if blen * copies * 2 > total:
            continue
        block = random_codes(rng, blen, gc)
        for _ in range(copies):
            ci = int(rng.integers(0, len(contigs)))
            c = contigs[ci]
            if len(c) <= blen + 2:
                continue
            pos = int(rng.integers(0, len(c) - blen))
            cp = mutate(block, rng, div, 0.0, 0.0)[:blen]
            if rng.random() < 0.5:
                cp = _COMP[cp[::-1]]
            c[pos:pos + len(cp)] = cp
    for _ in range(n_runs):
        ci = int(rng.integers(0, len(contigs)))
        c = contigs[ci]
        ln = int(rng.integers(50, 500))
        if len(c) > ln + 2:
            pos = int(rng.integers(0, len(c) - ln))
            c[pos:pos + ln] = 4
    return contigs


def codes_to_str(codes):
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    return lut[codes].tobytes().decode()


def read_lengths(rng, n, n50, sigma=0.75, lo=500, hi=100000):
    """log-normal lengths rescaled so that the sample N50 is ~n50"""
    ln = rng.lognormal(0.0, sigma, n)
    s = np.sort(ln)[::-1]
    cs = np.cumsum(s)
    cur_n50 = s[np.searchsorted(cs, cs[-1] / 2)]
    ln = ln * (n50 / cur_n50)
    return np.clip(ln, lo, hi).astype(np.int64)


def make_reads(seed, contigs, n_reads, n50=8000, sigma=0.75, lo=500, hi=100000, sub=0.024, ins=0.016, dele=0.020,
               lengths=None):
    """Returns (list of read strings, truth list of (contig, start, end, strand))."""
    rng = _rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    if lengths is None:
        lengths = read_lengths(rng, n_reads, n50, sigma, lo, hi)
    reads, truth = [], []
    p = lens / lens.sum()
    for i in range(n_reads):
        ci = int(rng.choice(len(contigs), p=p))
        L = int(min(lengths[i], lens[ci]))
        st = int(rng.integers(0, lens[ci] - L + 1))
        seg = contigs[ci][st:st + L]
        strand = 1
        if rng.random() < 0.5:
            strand = -1
            seg = np.where(seg < 4, 3 - seg, 4).astype(np.uint8)[::-1]
        rd = mutate(seg, rng, sub, ins, dele)
        reads.append(codes_to_str(rd))
        truth.append((ci, st, st + L, strand))
    return reads, truth


def write_fasta(path, contigs, names=None, width=80):
    with open(path, "w") as fh:
        for i, c in enumerate(contigs):
            fh.write(">%s\n" % (names[i] if names else "ctg%d" % i))
            s = codes_to_str(c)
            for j in range(0, len(s), width):
                fh.write(s[j:j + width] + "\n")


# canned configs (BASELINE.json configs[1] at reduced and full size)
def ecoli_like(seed=1, size=4641652):
    return make_genome(seed, [size], gc=0.508)


# ---------------------------------------------------------------------------------------------------------------
# GRCh38-scale synthetic genome (BASELINE.json configs[2]/[3]): 24 contigs with the hg38 primary-assembly
# lengths, i.i.d. background at GC 41 %, three injected repeat families and an N-run at every contig centre.
HG38_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
             135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
             46709983, 50818468, 156040895, 57227415]
HG38_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]


def _revcomp_rows(a):
    return (3 - a)[:, ::-1]


def _inject_family(rng, flat, starts, lens, consensus, n_copies, div, piece_len):
    """scatter n_copies diverged copies of the 3' `piece_len` bases of `consensus` at random places (vectorised)"""
    if n_copies <= 0:
        return
    total = int(flat.size)
    piece = consensus[-piece_len:]
    step = max(1000, 50_000_000 // piece_len)   # bounds the (m x piece_len) index array
    for lo in range(0, n_copies, step):
        m = min(step, n_copies - lo)
        pos = rng.integers(0, total - piece_len, m)
        # keep copies inside one contig
        ci = np.searchsorted(starts, pos, side="right") - 1
        ok = pos + piece_len <= starts[ci] + lens[ci]
        pos = pos[ok]
        m = len(pos)
        if m == 0:
            continue
        cp = np.broadcast_to(piece, (m, piece_len)).copy()
        mut = rng.random((m, piece_len), dtype=np.float32) < div
        cp[mut] = (cp[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) & 3
        flip = rng.random(m) < 0.5
        cp[flip] = _revcomp_rows(cp[flip])
        idx = pos[:, None] + np.arange(piece_len)[None, :]
        flat[idx.ravel()] = cp.ravel()


def make_human_like(seed=3, scale=1.0, log=None):
    """returns (list of contig code arrays [views of one flat array], names).  scale < 1 shrinks every contig."""
    rng = _rng(seed)
    lens = np.array([max(100000, int(l * scale)) for l in HG38_LENS], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    total = int(lens.sum())
    lut = np.empty(256, np.uint8)
    n_gc = int(round(256 * 0.41))
    lut[:n_gc] = np.where(np.arange(n_gc) % 2 == 0, 1, 2)             # C / G
    lut[n_gc:] = np.where(np.arange(256 - n_gc) % 2 == 0, 0, 3)       # A / T
    flat = np.empty(total, np.uint8)
    step = 1 << 28
    for lo in range(0, total, step):
        hi = min(total, lo + step)
        flat[lo:hi] = lut[rng.integers(0, 256, hi - lo, dtype=np.uint8)]
    if log: log("[synth] background %d bases" % total)
    sine = random_codes(rng, 300, 0.55)
    line = random_codes(rng, 6000, 0.40)
    sat = random_codes(rng, 171, 0.35)
    _inject_family(rng, flat, starts, lens, sine, int(1.1e6 * scale), 0.12, 300)
    for plen, frac in ((6000, 0.04), (3000, 0.08), (1500, 0.20), (900, 0.68)):
        _inject_family(rng, flat, starts, lens, line, int(0.5e6 * scale * frac), 0.08, plen)
    n_arr = max(1, int(total * 0.03 / 50000))
    for _ in range(n_arr):
        alen = 50000
        pos = int(rng.integers(0, total - alen))
        ci = int(np.searchsorted(starts, pos, side="right") - 1)
        if pos + alen > starts[ci] + lens[ci]:
            continue
        arr = np.tile(sat, alen // 171 + 1)[:alen].copy()
        mut = rng.random(alen, dtype=np.float32) < 0.02
        arr[mut] = (arr[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) & 3
        flat[pos:pos + alen] = arr
    for s, l in zip(starts, lens):                                    # centromere-like N-run
        nl = int(min(500000, l // 50))
        c = int(s + l // 2)
        flat[c:c + nl] = 4
    if log: log("[synth] repeats injected")
    contigs = [flat[int(s):int(s + l)] for s, l in zip(starts, lens)]
    return contigs, list(HG38_NAMES)


def make_reads_codes(seed, contigs, n_reads, n50=10000, sigma=0.75, lo=500, hi=100000, sub=0.024, ins=0.016, dele=0.020):
    """like make_reads but returns raw code bytes (0..4), which the C-ABI accepts like ASCII; avoids string conversion"""
    rng = _rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    lengths = read_lengths(rng, n_reads, n50, sigma, lo, hi)
    p = lens / lens.sum()
    cis = rng.choice(len(contigs), size=n_reads, p=p)
    out, truth = [], []
    for i in range(n_reads):
        ci = int(cis[i])
        L = int(min(lengths[i], lens[ci]))
        st = int(rng.integers(0, lens[ci] - L + 1))
        seg = contigs[ci][st:st + L]
        strand = 1
        if rng.random() < 0.5:
            strand = -1
            seg = np.where(seg < 4, 3 - seg, 4).astype(np.uint8)[::-1]
        out.append(mutate(seg, rng, sub, ins, dele).tobytes())
        truth.append((ci, st, st + L, strand))
    return out, truth


# ---------------------------------------------------------------------------------------------------------------
# ONE read set that any number of ranks can shard: the set is a sequence of fixed-size blocks, block b drawn from its own
# PCG64 stream (seed, b), so a rank can list the lengths of every read cheaply and synthesise only the blocks its shard overlaps.
READ_BLOCK = 4096


def block_lengths(seed, block, n50=10000, sigma=0.75, lo=500, hi=100000):
    """lengths of the READ_BLOCK reads of block `block` (the first draw of the block's stream)"""
    return read_lengths(_rng([int(seed), int(block)]), READ_BLOCK, n50, sigma, lo, hi)


def read_set_lengths(seed, n_reads, **kw):
    """lengths of reads 0..n_reads-1 of the read set `seed` (nominal lengths, before clipping to the contig and before indel errors)"""
    nb = (n_reads + READ_BLOCK - 1) // READ_BLOCK
    if nb == 0:
        return np.zeros(0, np.int64)
    return np.concatenate([block_lengths(seed, b, **kw) for b in range(nb)])[:n_reads]


def make_read_block(seed, block, contigs, n50=10000, sigma=0.75, lo=500, hi=100000, sub=0.024, ins=0.016, dele=0.020, as_codes=True):
    """the READ_BLOCK reads of block `block`: list of code bytes (0..4) or ASCII strings, plus the truth tuples"""
    rng = _rng([int(seed), int(block)])
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    lengths = read_lengths(rng, READ_BLOCK, n50, sigma, lo, hi)
    cis = rng.choice(len(contigs), size=READ_BLOCK, p=lens / lens.sum())
    out, truth = [], []
    for i in range(READ_BLOCK):
        ci = int(cis[i])
        L = int(min(lengths[i], lens[ci]))
        st = int(rng.integers(0, lens[ci] - L + 1))
        seg = contigs[ci][st:st + L]
        strand = 1
        if rng.random() < 0.5:
            strand = -1
            seg = np.where(seg < 4, 3 - seg, 4).astype(np.uint8)[::-1]
        rd = mutate(seg, rng, sub, ins, dele)
        out.append(rd.tobytes() if as_codes else codes_to_str(rd))
        truth.append((ci, st, st + L, strand))
    return out, truth


def read_set_slice(seed, lo_idx, hi_idx, contigs, **kw):
    """reads lo_idx..hi_idx-1 of the read set `seed` (only the overlapping blocks are synthesised)"""
    out = []
    for b in range(lo_idx // READ_BLOCK, (hi_idx + READ_BLOCK - 1) // READ_BLOCK):
        rd, _ = make_read_block(seed, b, contigs, **kw)
        b0 = b * READ_BLOCK
        out.extend(rd[max(lo_idx, b0) - b0:min(hi_idx, b0 + READ_BLOCK) - b0])
    return out
