#!/bin/bash
# Round-6 profiles, every BASELINE config from files of THIS round: rocprofv3 kernel stats (and, for the HBM-bound configs 3 and 5, the two
# PMC passes FETCH_SIZE / WRITE_SIZE: separate runs, nothing beside --pmc) of
#   config 2  bench.py default workload (25FV47, explicit carry, certificate inside the step)            -> kernel_stats_25fv47.csv
#   config 2  exact: tools/exact_roofline.py 25FV47 (the fused update), and its step split in both modes -> kernel_stats_exact_25fv47.csv, exact_profile_steps*.txt
#   config 3  bench.py --workload dense4096 --dense-storage f64                                          -> kernel_stats_dense4096_f64.csv, pmc_traffic_dense4096_f64.json
#   config 5  bench.py --workload maxflow --crash 0 / --crash 1                                          -> kernel_stats_maxflow_*.csv, pmc_traffic_maxflow_*.json
# then the driver's own invocation (default bench line + bench_configs.json).  Run on the GPU box from the repo root; the summaries land in
# gpurun_out/prof_r6 and are copied into profiles/r6_* by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-configs --no-concurrency-probe"
stats() {  # name, program and arguments
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- "$@" > $OUT/${name}_under_rocprof.txt 2> $OUT/stats_$name.log
  find $OUT/stats_$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$name.csv \;
  rm -rf $OUT/stats_$name
}
pmc() {  # name, program and arguments
  local name=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$name -- "$@" > /dev/null 2> $OUT/fetch_$name.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$name -- "$@" > /dev/null 2> $OUT/write_$name.log
  python3 $R/tools/pmc_traffic.py $OUT/fetch_$name $OUT/write_$name $OUT/pmc_traffic_$name.json > $OUT/pmc_traffic_$name.txt 2>&1
  rm -rf $OUT/fetch_$name $OUT/write_$name
}
stats 25fv47 python3 $R/bench.py --steps 3 --warmup 1 $COMMON
stats exact_25fv47 python3 $R/tools/exact_roofline.py 25FV47
stats dense4096_f64 python3 $R/bench.py --steps 3 --warmup 1 $COMMON --workload dense4096 --dense-storage f64
pmc dense4096_f64 python3 $R/bench.py --steps 1 --warmup 0 $COMMON --workload dense4096 --dense-storage f64
# (no PMC pass for the exact solve: rocprofv3 --pmc segfaults behind its cooperative launches on this image; its roofline's `traffic` stays null)
MF="--steps 1 --warmup 0 $COMMON --workload maxflow"
stats maxflow_reference_start python3 $R/bench.py $MF --crash 0
pmc maxflow_reference_start python3 $R/bench.py $MF --crash 0
stats maxflow_crash python3 $R/bench.py $MF --crash 1
pmc maxflow_crash python3 $R/bench.py $MF --crash 1
cd $R
for mode in 0 4; do
  RELP_EXACT_UPDATE=$mode RELP_EXACT_PROFILE=1 python3 tools/exact_roofline.py 25FV47 E226 > $OUT/exact_roofline_mode$mode.txt 2> $OUT/exact_profile_steps_mode$mode.txt
done
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default_stdout.txt 2> $OUT/bench_default.err   # the driver's invocation
tail -1 $OUT/bench_default_stdout.txt > $OUT/bench_default_line.json
cp bench_configs.json $OUT/bench_configs.json
ls -la $OUT
