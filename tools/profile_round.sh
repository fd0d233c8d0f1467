#!/bin/bash
# the round's evidence: default bench line, rocprofv3 kernel statistics of the same workload, HBM traffic (PMC) of one sub-batch.
# usage (on the GPU box, from the repo root): tools/profile_round.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -o r04 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-resident > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o r04f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-pcie --streams 1 --depth 1 --reads 9216 --synth-procs 1 > $out/bench_pmc_fetch.json 2> $out/bench_pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o r04w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-pcie --streams 1 --depth 1 --reads 9216 --synth-procs 1 > $out/bench_pmc_write.json 2> $out/bench_pmc_write.err
cd $GRAFT_REPO_ROOT
python tools/pmcsum.py $(find $out/pmc_fetch -name "*counter_collection.csv") > $out/pmc_fetch_by_kernel.txt 2>&1
python tools/pmcsum.py $(find $out/pmc_write -name "*counter_collection.csv") > $out/pmc_write_by_kernel.txt 2>&1
find $out -name "*counter_collection.csv" -size +20M -delete
find $out -name "*kernel_trace.csv" -size +30M -delete
head -c 600 $out/bench_default.json; echo; head -12 $out/pmc_fetch_by_kernel.txt
python tools/pmc2json.py human/9216 $out/pmc_fetch_by_kernel.txt $out/pmc_write_by_kernel.txt $out/pmc_traffic.json
