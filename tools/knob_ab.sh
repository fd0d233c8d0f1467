#!/bin/bash
# A/B of scheduling knobs on the default workload, 8 steps each: tools/knob_ab.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; "$@" > $out/$tag.json 2> $out/$tag.err; python3 - $out/$tag.json $tag <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'])
P
}
run base1 python3 bench.py --no-cpu --no-pcie --steps 8 --warmup 1
MM355_DP_TURNS=2 run turns2 python3 bench.py --no-cpu --no-pcie --steps 8 --warmup 1
run streams7 python3 bench.py --no-cpu --no-pcie --steps 8 --warmup 1 --streams 7
run base2 python3 bench.py --no-cpu --no-pcie --steps 8 --warmup 1
