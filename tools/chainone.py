import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mappy-rs_amd"))
import numpy as np, synthdata as S, mappy_rs
g = S.ecoli_like(1); S.write_fasta("/tmp/cp.fa", g)
reads, _ = S.make_reads(2, g, 16384, n50=8000)
al = mappy_rs.Aligner("/tmp/cp.fa", preset="map-ont"); sr = al._stage_runner()
lens = np.array([len(r) for r in reads]); r = reads[int(np.argmax(lens))]
for rep in range(3): sr.chains([r], cap=4_000_000)
st = sr.stats(); print("chain %.2f ms n_a %d pairs %d" % (st.ms_chain, st.n_a, st.chain_pairs))
