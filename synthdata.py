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
