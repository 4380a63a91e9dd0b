#!/bin/bash
# Round-2 profile of BASELINE config 3 (dense 4096 x 8192): rocprofv3 kernel stats of bench.py --workload dense4096, then the
# bench lines with the block as signed bytes (narrowest exact type), as float and as double.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r2_dense
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --workload dense4096 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_dense4096.csv \;
rm -rf $OUT/stats
cd $R
python3 bench.py --workload dense4096 --no-cpu-baseline > $OUT/bench_dense4096_i8.json 2> /dev/null
python3 bench.py --workload dense4096 --no-cpu-baseline --dense-storage f32 > $OUT/bench_dense4096_f32.json 2> /dev/null
python3 bench.py --workload dense4096 --no-cpu-baseline --dense-storage f64 > $OUT/bench_dense4096_f64.json 2> /dev/null
python3 bench.py --workload netlib > $OUT/bench_netlib.json 2> /dev/null
ls -la $OUT
