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
# HBM traffic of the same workload: FETCH_SIZE and WRITE_SIZE in separate passes (no trace domains beside --pmc)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --workload dense4096 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --workload dense4096 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/write.log
python3 $R/tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic_dense4096.json > $OUT/pmc_traffic_dense4096.txt 2>&1
rm -rf $OUT/fetch $OUT/write
cd $R
python3 bench.py --workload dense4096 --no-cpu-baseline > $OUT/bench_dense4096_i8.json 2> /dev/null
python3 bench.py --workload dense4096 --no-cpu-baseline --dense-storage f32 > $OUT/bench_dense4096_f32.json 2> /dev/null
python3 bench.py --workload dense4096 --no-cpu-baseline --dense-storage f64 > $OUT/bench_dense4096_f64.json 2> /dev/null
python3 bench.py --workload netlib > $OUT/bench_netlib.json 2> /dev/null
ls -la $OUT
