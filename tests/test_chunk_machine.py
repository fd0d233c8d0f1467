"""The chunk machine of the sketch kernels (mappy-rs_amd/csrc/mm355_sketch.h: a lane sketches 384 bases of a read behind a warm-up whose
sufficiency it PROVES while running, plain and homopolymer-compressed) compiled for the host (tests/host_harness/chunk_sketch_host.cpp, g++)
and held against the oracle's sequential U:sketch.c::mm_sketch on the inputs that stress the proof -- the CPU counterpart of
tests/test_gpu_stages.py::test_chunked_sketch_adversarial and tests/test_gpu_hpc.py::test_hpc_chunked_sketch_parity (same header, no GPU)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
import synthdata as S

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def chunk_lib(built):
    src = os.path.join(HERE, "host_harness", "chunk_sketch_host.cpp")
    so = os.path.join(HERE, "host_harness", "libchunkhost.so")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "..", "mappy-rs_amd", "csrc", "mm355_sketch.h"))):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-o", so, src])
    L = C.CDLL(so)
    L.chunk_sketch_host.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.chunk_sketch_host.restype = C.c_int
    return L


def _oracle_sketch(b, w, k, hpc):
    v = O.MM128V()
    O.lib().mmo_sketch(b, len(b), w, k, 0, hpc, C.byref(v))
    out = np.ctypeslib.as_array(C.cast(v.a, C.POINTER(C.c_uint64)), shape=(v.n, 2)).copy() if v.n else np.zeros((0, 2), np.uint64)
    O.lib().free(v.a)
    return out


def _reads():
    from test_host import _hp_genome
    rng = np.random.default_rng(13)
    g = _hp_genome(73, [30000], repeats=())
    base = S.codes_to_str(g[0][:14000])
    plain = S.codes_to_str(S.random_codes(rng, 4000))
    long_runs = plain[:700] + "A" * 380 + plain[700:1100] + "C" * 800 + plain[1100:1500] + "G" * 255 + plain[1500:1900] + "T" * 256 + plain[1900:]
    with_n = "".join(c if i % 11 else "N" for i, c in enumerate(base[:5000]))
    n_in_run = plain[:900] + "AAAAANAAAAA" + plain[900:1300] + "N" * 400 + "CCCC" + plain[1300:2500]
    pal = "ACGT" * 300 + "AT" * 500 + "GATC" * 200 + "AACCGGTT" * 150
    runs = plain[:1000] + "N" * 700 + plain[1000:1400] + "N" * 33 + plain[1400:3000]
    return [base, plain, long_runs, with_n, n_in_run, pal, runs, "A" * 3000, "AC" * 1500, "N" * 700 + base[:900], base[:383], base[:384], base[:385],
            base[:769], "A" * 384 + plain[:500], plain[:380] + "T" * 10 + plain[380:900], S.codes_to_str(S.random_codes(rng, 5000, gc=0.08)), "ACGT", "A", "N" * 50]


@pytest.mark.parametrize("hpc", [0, 1])
@pytest.mark.parametrize("piece", [0, 32])        # a lane per 384-base chunk (k_sketch) / per 32-base piece (k_sketch_sparse)
def test_chunk_machine_equals_the_sequential_sketch(chunk_lib, hpc, piece):
    reads = _reads()
    n = 0
    for k, w in ((15, 10), (19, 10), (19, 5), (14, 8), (16, 5), (21, 11), (19, 19), (28, 30)):
        for i, rd in enumerate(reads):
            b = rd.encode()
            exp = _oracle_sketch(b, w, k, hpc)
            out = np.zeros((len(b) + 8, 2), np.uint64)
            m = chunk_lib.chunk_sketch_host(b, len(b), w, k, hpc, piece, out.ctypes.data)
            assert m == len(exp) and np.array_equal(out[:m], exp), (k, w, i, m, len(exp))
            n += m
    assert n > 20000


def test_chunk_machine_on_random_sequences(chunk_lib):
    """random sequences (plain, run-rich, low-complexity periodic, N-rich, runs of hundreds of bases) x random k (5..28), w (1..63), plain / HPC,
    chunk sizes from 17 to 1000 bases (the proof must not depend on where a chunk starts); 3000 cases of the same generator ran clean while
    the HPC machine was written (1.9 M minimizers)"""
    rng = np.random.default_rng(2)
    tot = 0
    for it in range(400):
        n = int(rng.integers(1, 5000))
        mode = int(rng.integers(0, 5))
        if mode == 0:
            codes = rng.integers(0, 4, n)
        elif mode == 1:
            codes = np.repeat(rng.integers(0, 4, n), np.where(rng.random(n) < 0.2, rng.integers(2, 60, n), 1))[:n]
        elif mode == 2:
            p = int(rng.integers(1, 7))
            codes = np.tile(rng.integers(0, 4, p), n // p + 1)[:n].copy()
            m = rng.random(n) < 0.02
            codes[m] = rng.integers(0, 4, int(m.sum()))
        elif mode == 3:
            codes = rng.integers(0, 4, n)
            codes[rng.random(n) < rng.choice([0.002, 0.02, 0.2])] = 4
            if rng.random() < 0.5:
                a = int(rng.integers(0, n)); codes[a:a + int(rng.integers(1, 800))] = 4
        else:
            codes = np.repeat(rng.integers(0, 4, n), np.where(rng.random(n) < 0.01, rng.integers(100, 900, n), 1))[:n]
        b = bytes(b"ACGTN"[int(x)] for x in codes)
        k, w, hpc = int(rng.integers(5, 29)), int(rng.integers(1, 64)), int(rng.integers(0, 2))
        piece = int(rng.choice([0, 32, 17, 100, 384, 1000]))
        exp = _oracle_sketch(b, w, k, hpc)
        out = np.zeros((len(b) + 8, 2), np.uint64)
        m = chunk_lib.chunk_sketch_host(b, len(b), w, k, hpc, piece, out.ctypes.data)
        assert m == len(exp) and np.array_equal(out[:m], exp), (it, n, mode, k, w, hpc, piece, m, len(exp))
        tot += m
    assert tot > 100000
