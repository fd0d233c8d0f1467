#!/bin/bash
# host-side phase timeline (MM355_TRACE) of the bench: tools/trace_profile.sh <outdir under gpurun_out> [bench.py arguments ...]
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out; shift
cd $GRAFT_REPO_ROOT
MM355_TRACE=$out/trace.tsv timeout 500 python3 bench.py --no-cpu --no-pcie "$@" > $out/bench.json 2> $out/bench.err
python3 tools/tracesum.py $out/trace.tsv > $out/tracesum.txt 2>&1
cat $out/tracesum.txt; head -c 300 $out/bench.json
