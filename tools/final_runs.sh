#!/bin/bash
# the round's closing evidence after tools/profile_round.sh: the driver's command, the other workloads, latency table, python surface
# usage (on the GPU box, from the repo root): tools/final_runs.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_command.json 2> $out/bench_driver_command.err
timeout 400 python3 bench.py --workload ecoli --cpu-seconds 10 > $out/bench_ecoli.json 2> $out/bench_ecoli.err
timeout 400 python3 bench.py --workload human-hifi --cpu-seconds 10 > $out/bench_human_hifi.json 2> $out/bench_human_hifi.err
timeout 400 python3 bench.py --workload ecoli-hifi --cpu-seconds 10 > $out/bench_ecoli_hifi.json 2> $out/bench_ecoli_hifi.err
timeout 300 python3 tools/latency_table.py > $out/latency_table.txt 2> $out/latency_table.err
timeout 400 python3 tools/mapbatch_bench.py 262144 8 > $out/mapbatch.txt 2> $out/mapbatch.err
for f in bench_driver_command bench_ecoli bench_human_hifi bench_ecoli_hifi; do python3 - $out/$f.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], (d.get('cpu_baseline') or {}).get('parity'))
P
done
tail -12 $out/latency_table.txt; tail -5 $out/mapbatch.txt
