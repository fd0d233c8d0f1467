#!/bin/bash
# A/B sweep of bench.py variants on the GPU box: tools/sweep.sh <outdir> [variant-file]
# variant file: one variant per line, "name|ENV=.. ENV=..|bench args"; default = the stream / hardware-queue variants below
out=$1; mkdir -p $out
run() { name=$1; envs=$2; args=$3; echo "== $name: $envs bench.py $args" >> $out/sweep.log
  env $envs timeout 600 python bench.py --no-cpu --no-resident --steps 3 --warmup 1 $args > $out/$name.json 2> $out/$name.err
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
  run base "A=1" ""
  run dpstreams0 "MM355_DP_STREAMS=0" ""
  run q16 "GPU_MAX_HW_QUEUES=16" ""
  run q16_dps0 "GPU_MAX_HW_QUEUES=16 MM355_DP_STREAMS=0" ""
  run q24_dps0 "GPU_MAX_HW_QUEUES=24 MM355_DP_STREAMS=0" ""
  run q24_dps0_glob "GPU_MAX_HW_QUEUES=24 MM355_DP_STREAMS=0 MM355_DP_GLOBAL_STREAMS=1" ""
  run s8d3 "A=1" "--streams 8 --depth 3"
  run s12d2 "A=1" "--streams 12 --depth 2"
  run q24_dps0_s12d2 "GPU_MAX_HW_QUEUES=24 MM355_DP_STREAMS=0" "--streams 12 --depth 2"
  run turns2 "MM355_DP_TURNS=2" ""
fi
cat $out/sweep.log
