#!/bin/bash
# Round-4 profile of the fixed-width exact simplex on the metric's LP: rocprofv3 kernel stats of `tools/exact_probe.py 25FV47`
# (relp_solve_exact, 4 -> 128 limbs: one cooperative launch of exact_simplex_kernel<L> per width).  Run on the GPU box from the repo
# root; the summary lands in gpurun_out/prof_r4_exact and is copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r4_exact
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RELP_PROBE_FIRST_LIMBS=4 RELP_PROBE_MAX_LIMBS=128
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/exact_probe.py 25FV47 > $OUT/exact_probe_under_rocprof.txt 2> $OUT/stats.log
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_exact_25fv47.csv \;
rm -rf $OUT/stats
ls -la $OUT
