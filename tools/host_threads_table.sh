#!/bin/bash
# Mbases/s against the size of the library's host pool (MM355_HOST_THREADS): what one rank of eight needs of a node's cores
# usage (GPU box, repo root): tools/host_threads_table.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
echo "MM355_HOST_THREADS  Mbases/s(PCIe-inclusive)  ms/step  host_cpu_us_per_read  pool_busy_frac" > $out/table.txt
for t in 4 6 8 10 14; do
  MM355_HOST_THREADS=$t timeout 400 python3 bench.py --no-cpu --no-resident --steps 8 > $out/h$t.json 2> $out/h$t.err
  python3 - >> $out/table.txt <<PY
import json
d=json.loads(open("$out/h$t.json").read().strip().splitlines()[-1])
print("%18d  %24.1f  %7.1f  %20.1f  %14.3f" % ($t, d["value"], d["ms_per_step"], d["host"]["cpu_us_per_read"], d["host"]["busy_frac_of_pool"]))
PY
done
cat $out/table.txt
