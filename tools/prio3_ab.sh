#!/bin/bash
# A/B of the three priority levels of the extension streams (MM355_DP_PRIO3, mm355_pipeline.hip::dp_stream_prio) on the default workload,
# alternating runs on one box + MM355_TRACE medians of the turn: tools/prio3_ab.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; "$@" > $out/$tag.json 2> $out/$tag.err; python3 - $out/$tag.json $tag <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'])
P
}
for i in 1 2 3; do
MM355_DP_QALIGN=1 run qal$i python3 bench.py --no-cpu --no-resident --steps 10 --warmup 1
MM355_DP_PRIO3=0 run off$i python3 bench.py --no-cpu --no-resident --steps 10 --warmup 1
done
