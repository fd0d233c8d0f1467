#!/bin/bash
# kernel statistics of ONE sub-batch alone on the GPU (no other context running): tools/alone_profile.sh <outdir under gpurun_out> [reads]
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o al -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-pcie --streams 1 --depth 1 --reads ${2:-9216} --synth-procs 1 > $out/bench.json 2> $out/bench.err
find $out -name "*kernel_trace.csv" -size +30M -delete
tail -c 400 $out/bench.json
