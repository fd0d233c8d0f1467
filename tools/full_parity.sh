#!/bin/bash
# whole-block parity: a CPU leg long enough for the oracle to map every read of a block; tools/full_parity.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 700 python3 bench.py --steps 2 --no-pcie --cpu-seconds 200 > $out/human.json 2> $out/human.err
timeout 500 python3 bench.py --workload human-hifi --steps 2 --no-pcie --cpu-seconds 120 > $out/human_hifi.json 2> $out/human_hifi.err
timeout 400 python3 bench.py --workload ecoli-hifi --steps 2 --no-pcie --cpu-seconds 60 > $out/ecoli_hifi.json 2> $out/ecoli_hifi.err
for f in human human_hifi ecoli_hifi; do python3 - $out/$f.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['value'], d['cpu_baseline']['parity'], d['cpu_baseline']['value'])
P
done
