"""chain-stage timing on the default (E. coli-like) workload: whole batch, the longest read alone, the 64 longest reads"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mappy-rs_amd"))
import numpy as np
import synthdata as S
import mappy_rs
from mappy_rs import _ffi
g = S.ecoli_like(1)
S.write_fasta("/tmp/cp.fa", g)
reads, _ = S.make_reads(2, g, 16384, n50=8000)
al = mappy_rs.Aligner("/tmp/cp.fa", preset="map-ont")
sr = al._stage_runner()
lens = np.array([len(r) for r in reads]); order = np.argsort(-lens)
for name, sel in (("all 16384", list(range(len(reads)))), ("longest", [order[0]]), ("64 longest", list(order[:64])), ("64 median", list(order[8000:8064]))):
    rr = [reads[i] for i in sel]
    for rep in range(2):
        sr.chains(rr, cap=40_000_000); st = sr.stats()
    print("%-12s reads %5d max_len %6d: sort %.2f chain %.2f backtrack %.2f ms  (n_a %d, pairs %d)" % (name, len(rr), max(len(x) for x in rr), st.ms_sort, st.ms_chain, st.ms_backtrack, st.n_a, st.chain_pairs))
