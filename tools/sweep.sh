#!/bin/bash
# A/B sweep of bench.py variants on the GPU box: tools/sweep.sh <outdir> [variant-file]
# variant file: one variant per line, "name|ENV=.. ENV=..|bench args"; default = the variants below (the first and the last are the same: their difference is the noise of the box)
out=$1; mkdir -p $out
run() { name=$1; envs=$2; args=$3; echo "== $name: $envs bench.py $args" >> $out/sweep.log
  env $envs timeout 600 python bench.py --no-cpu --no-pcie --steps 6 --warmup 1 $args > $out/$name.json 2> $out/$name.err
  python - "$out/$name.json" >> $out/sweep.log <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("   value %.1f Mbases/s  ms/step %.1f  kernel_ms %s" % (d['value'], d['ms_per_step'], d['kernel_ms_per_step']))
except Exception as e: print("   FAILED", e)
PY
}
if [ -n "$2" ]; then
  while IFS='|' read -r name envs args; do [ -n "$name" ] && run "$name" "${envs:-A=1}" "$args"; done < "$2"
else
  # (every knob below exists in the library: MM355_DP_TURNS, MM355_DP_REGW8, MM355_DP_SHARED_STREAMS, MM355_RMQ_ON_HOST, MM355_BUF_SLACK_DIV, GPU_MAX_HW_QUEUES)
  run base "A=1" ""
  run q16 "GPU_MAX_HW_QUEUES=16" ""
  run q4 "GPU_MAX_HW_QUEUES=4" ""
  run turns2 "MM355_DP_TURNS=2" ""
  run regw1 "MM355_DP_REGW8=0" ""
  run shared "MM355_DP_SHARED_STREAMS=1" ""
  run rmqhost "MM355_RMQ_ON_HOST=1" ""
  run bin "A=1" "--bin"
  run s8d2 "MM355_BUF_SLACK_DIV=8" "--reads 98304 --streams 8 --depth 2"
  run base2 "A=1" ""
fi
cat $out/sweep.log
