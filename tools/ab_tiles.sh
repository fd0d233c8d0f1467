#!/bin/bash
# A/B of two builds of libmm355.so on the extension stage alone: kernel statistics (rocprofv3 --kernel-trace --stats) and FETCH_SIZE / WRITE_SIZE
# of tools/dpbench.py for each library.  usage (on the GPU box, from the repo root): tools/ab_tiles.sh <outdir under gpurun_out> <old .so> [L n]
out=$GRAFT_REPO_ROOT/$1; mkdir -p $out
old=$GRAFT_REPO_ROOT/$2
L=${3:-210}; n=${4:-100000}
new=$GRAFT_REPO_ROOT/mappy-rs_amd/csrc/libmm355.so
cd /tmp; export TMPDIR=/tmp
for tag in old new; do
	if [ $tag = old ]; then export MM355_LIB_PATH=$old; else export MM355_LIB_PATH=$new; fi
	timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$tag -o ks -- python3 $GRAFT_REPO_ROOT/tools/dpbench.py $L $n 8 30001 > $out/dpbench_$tag.txt 2>&1
	timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf_$tag -o pf -- python3 $GRAFT_REPO_ROOT/tools/dpbench.py $L $n 8 30001 > /dev/null 2>&1
	timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw_$tag -o pw -- python3 $GRAFT_REPO_ROOT/tools/dpbench.py $L $n 8 30001 > /dev/null 2>&1
	python3 $GRAFT_REPO_ROOT/tools/pmcsum.py $(find $out/pf_$tag $out/pw_$tag -name "*counter_collection.csv") > $out/pmc_$tag.txt 2>&1
	find $out -name "*counter_collection.csv" -size +5M -delete
	find $out -name "*kernel_trace.csv" -size +5M -delete
	echo "== $tag"; tail -3 $out/dpbench_$tag.txt
	grep -h "k_ksw" $(find $out/ks_$tag -name "*kernel_stats.csv") | cut -d, -f1-4 | sed 's/(DpConst[^"]*"/"/; s/(DpJobDev[^"]*"/"/'
	grep "k_ksw" $out/pmc_$tag.txt
done
