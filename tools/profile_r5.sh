#!/bin/bash
# Round-5 profiles: rocprofv3 kernel stats of the headline bench command and of the exact solve of 25FV47 (the update of N on the
# matrix cores), the exact path's step seconds / word-product counts, the int-mul micro-benchmark.  Run on the GPU box from the repo
# root; the summaries land in gpurun_out/prof_r5 and are copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-configs --no-concurrency-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_25fv47 -- python3 $R/bench.py --steps 3 --warmup 1 $COMMON > $OUT/bench_25fv47_under_rocprof.json 2> $OUT/stats_25fv47.log
find $OUT/stats_25fv47 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_25fv47.csv \;
rm -rf $OUT/stats_25fv47
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_exact -- python3 $R/tools/exact_roofline.py 25FV47 > $OUT/exact_roofline_under_rocprof.txt 2> $OUT/stats_exact.log
find $OUT/stats_exact -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_exact_25fv47.csv \;
rm -rf $OUT/stats_exact
cd $R
RELP_EXACT_PROFILE=1 python3 tools/exact_roofline.py 25FV47 E226 > $OUT/exact_roofline.txt 2> $OUT/exact_profile_stderr.txt
ls -la $OUT
