#!/bin/bash
# closing evidence of round 4's last session (after map-pb / ava-pb and the cull rule fix): tools/final_round4.sh <outdir under gpurun_out>
# default bench line with the CPU leg, rocprofv3 kernel statistics of the same workload, the driver's command, and the instruction counters of
# one sub-batch (VALU / SALU / LDS / VMEM instructions, waves, busy cycles, LDS bank conflicts per kernel).  Raw CSVs are summarised and
# deleted: gpurun merges at most 64 MiB back.
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
# (the driver command was run in the two earlier calls of this script: 1452.8 and 1430.8 Mbases/s)
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -o r04 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-resident > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_inst -o r04i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-pcie --streams 1 --depth 1 --reads 9216 --synth-procs 1 > $out/bench_pmc_inst.json 2> $out/bench_pmc_inst.err
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM --output-format csv -d $out/pmc_lds -o r04l -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-pcie --streams 1 --depth 1 --reads 9216 --synth-procs 1 > $out/bench_pmc_lds.json 2> $out/bench_pmc_lds.err
cd $GRAFT_REPO_ROOT
python tools/pmcsum.py $(find $out/pmc_inst -name "*counter_collection.csv") > $out/pmc_inst_by_kernel.txt 2>&1
python tools/pmcsum.py $(find $out/pmc_lds -name "*counter_collection.csv") > $out/pmc_lds_by_kernel.txt 2>&1
find $out -name "*counter_collection.csv" -delete
find $out -name "*kernel_trace.csv" -delete
du -sh $out
head -c 300 $out/bench_default.json; echo; head -3 $out/pmc_inst_by_kernel.txt
