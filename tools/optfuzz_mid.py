"""option fuzzing on an ANCHOR-RICH, SPARSE reference (the 155-Mbp human-like genome of the mid-scale tests): the cull + LDS sort + literal
tie path is what every read takes there, and lone repeat hits are what the cull drops -- the small genome of tools/optfuzz.py has no empty
position bins, so a hole in the cull's rule cannot show on it (the min_cnt term of round 4's T was found on an HPC index, not by that fuzzer).
One device index and one oracle index per preset; every configuration changes MAP options only (chaining and DP thresholds, band widths,
gaps, scoring, flags), on both option structs by field name; 28 reads + 4 chimeras per configuration, every hit record compared.
python tools/optfuzz_mid.py [seed=1] [n_configs=24] [preset=map-ont]   (needs the GPU)"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "mappy-rs_amd"))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
# NO_LJOIN, NO_END_FLT, HARD_MLEVEL, NO_INV, FOR_ONLY, REV_ONLY  (not ALL_CHAINS: thousands of chains per read on this genome, minutes of
# scalar ksw2 per read in the oracle; tools/optfuzz.py covers it on the small genome)
FLAGS = (0x400, 0x10000000, 0x20000000, 0x200000000, 0x100000, 0x200000)


def random_fields(rng, base):
    f = {}
    if rng.random() < 0.6: f["min_cnt"] = int(rng.integers(1, 7))
    if rng.random() < 0.7: f["min_chain_score"] = int(rng.integers(8, 130))
    if rng.random() < 0.4: f["min_dp_max"] = int(rng.integers(20, 300))
    if rng.random() < 0.5:
        f["bw"] = int(rng.choice([50, 100, 250, 500, 1000, 3000]))
        f["bw_long"] = max(f["bw"], int(rng.choice([500, 2000, 20000])))
    if rng.random() < 0.4: f["max_gap"] = int(rng.choice([1000, 2500, 5000, 10000]))
    if rng.random() < 0.2: f["max_gap_ref"] = int(rng.choice([3000, 8000]))
    if rng.random() < 0.3: f["max_frag_len"] = int(rng.choice([800, 5000, 20000]))
    if rng.random() < 0.4: f["best_n"] = int(rng.integers(1, 12))
    if rng.random() < 0.3: f["max_chain_skip"] = int(rng.choice([5, 25, 60]))
    if rng.random() < 0.3: f["max_chain_iter"] = int(rng.choice([200, 1000, 5000, 8000]))
    if rng.random() < 0.3: f["mid_occ"] = int(rng.choice([20, 60, 300, 1000]))
    if rng.random() < 0.2: f["occ_dist"] = int(rng.choice([0, 100, 2000]))
    if rng.random() < 0.2: f["max_max_occ"] = int(rng.choice([200, 1000, 4095]))
    if rng.random() < 0.2: f["rmq_inner_dist"] = int(rng.choice([200, 1000, 3000]))
    if rng.random() < 0.2: f["zdrop"] = int(rng.choice([100, 200, 400, 800])); f["zdrop_inv"] = min(f.get("zdrop", 400), int(rng.choice([100, 200])))
    if rng.random() < 0.5:
        a, b, q, e = int(rng.integers(1, 5)), int(rng.integers(1, 10)), int(rng.integers(1, 12)), int(rng.integers(1, 4))
        q2, e2 = int(rng.integers(q, 40)), int(rng.integers(1, e + 1))
        lim = 2 * min(q + e, q2 + e2)            # ksw2's domain (the product refuses what lies beyond it)
        f.update(a=a, b=min(b, lim), q=q, e=e, q2=q2, e2=e2)
        if f.get("min_dp_max") is None: f["min_dp_max"] = base.min_chain_score * a
    fl = 0
    for x in FLAGS:
        if rng.random() < 0.15: fl |= x
    if fl: f["flag"] = base.flag | fl
    return f


def run(seed, n_configs, preset):
    import mappy_rs
    from mappy_rs import _ffi
    import synthdata as S
    from oracle import oracle as O
    from test_gpu_human import build_device_index, ONT, HIFI
    L = _ffi.lib()
    rng = np.random.default_rng(seed)
    g, names = S.make_human_like(3, 0.05)
    kw = ONT if preset == "map-ont" else HIFI
    idx, mo0 = build_device_index(L, _ffi, g, names, preset)
    orc = O.OracleAligner(codes=g, names=names, preset=preset, n_threads=16)
    omo0 = bytes(orc.mo)
    comp = lambda c: np.where(c < 4, 3 - c, 4).astype(np.uint8)[::-1]
    ctx = C.c_void_p()
    _ffi.check(L.mm355_ctx_create(idx, 0, C.byref(ctx)))
    tot_hits = tot_bad = tot_culled = tot_a = 0
    for ci in range(n_configs):
        f = random_fields(rng, mo0)
        mo = _ffi.MapOpt.from_buffer_copy(bytes(mo0))
        C.memmove(C.byref(orc.mo), omo0, len(omo0))
        for k, v in f.items():
            setattr(mo, k, v); setattr(orc.mo, k, v)
        reads, _ = S.make_read_block(int(rng.integers(10, 1 << 20)), 0, g, **kw)
        reads = reads[:28]
        for _ in range(4):
            c0, c1 = g[int(rng.integers(0, len(g)))], g[int(rng.integers(0, len(g)))]
            a0, b0 = int(rng.integers(0, len(c0) - 8000)), int(rng.integers(0, len(c1) - 8000))
            c = np.concatenate([c0[a0:a0 + 3000], comp(c1[b0:b0 + 2500]), c0[a0 + 5000:a0 + 7500]])
            reads.append(S.mutate(c, rng, 0.03, 0.01, 0.01).tobytes())
        rarr, rlens, keep = _ffi.pack_reads(reads)
        hp = C.POINTER(_ffi.Hits)()
        rc = L.mm355_map_batch(ctx, C.byref(mo), len(reads), rarr, rlens, 1, C.byref(hp))
        if rc != 0:
            print("cfg %2d refused (rc %d) %s" % (ci, rc, f), flush=True)
            continue
        got = mappy_rs._batch_to_mappings(hp, len(reads), names)
        L.mm355_free_hits(hp)
        st = _ffi.Stats(); L.mm355_get_stats(ctx, C.byref(st))
        bad = nh = 0
        for i, rd in enumerate(reads):
            exp = orc.map(rd, cs=True)
            nh += len(exp)
            ok = len(exp) == len(got[i]) and all(
                (m.target_name, m.target_start, m.target_end, m.query_start, m.query_end, m.strand, m.mapq, m.is_primary, m.NM, m.match_len, m.block_len, m.cigar_str, m.cs) ==
                (e["target_name"], e["target_start"], e["target_end"], e["query_start"], e["query_end"], e["strand"], e["mapq"], e["is_primary"], e["NM"], e["match_len"],
                 e["block_len"], e["cigar_str"], e["cs"]) for m, e in zip(got[i], exp))
            bad += not ok
        tot_hits += nh; tot_bad += bad
        print("cfg %2d hits %4d mismatching reads %d  anchors kept %d of %d (this batch)  %s" % (ci, nh, bad, st.n_a_kept, st.n_a, f), flush=True)
        tot_culled, tot_a = st.n_a - st.n_a_kept, st.n_a
    L.mm355_ctx_destroy(ctx)
    L.mm355_index_free(idx)
    return n_configs, tot_hits, tot_bad, tot_culled, tot_a


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    preset = sys.argv[3] if len(sys.argv) > 3 else "map-ont"
    nc, nh, nb, ncul, na = run(seed, n, preset)
    print("preset %s seed %d: configs %d, hits %d, mismatching reads %d; anchors culled %d of %d" % (preset, seed, nc, nh, nb, ncul, na))
    sys.exit(1 if nb else 0)
