#!/bin/bash
# Round-4 profiles: rocprofv3 kernel stats and the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no trace domain beside
# --pmc) of bench.py for the explicit carry and the inverse-factor carry (refactorisation on the host and as kernels on the device) on 25FV47 and for the dense LP of config 3 with the block as double;
# then the default bench line (every BASELINE config) and the carry table.  Run on the GPU box from the repo root; the summaries
# land in gpurun_out/prof_r4 and are copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-configs --no-concurrency-probe"
profile() {  # name, bench arguments
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 $R/bench.py --steps 3 --warmup 1 $COMMON "$@" > $OUT/bench_${name}_under_rocprof.json 2> $OUT/stats_$name.log
  find $OUT/stats_$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$name.csv \;
  rm -rf $OUT/stats_$name
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$name -- python3 $R/bench.py --steps 1 --warmup 0 $COMMON "$@" > /dev/null 2> $OUT/fetch_$name.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$name -- python3 $R/bench.py --steps 1 --warmup 0 $COMMON "$@" > /dev/null 2> $OUT/write_$name.log
  python3 $R/tools/pmc_traffic.py $OUT/fetch_$name $OUT/write_$name $OUT/pmc_traffic_$name.json > $OUT/pmc_traffic_$name.txt 2>&1
  rm -rf $OUT/fetch_$name $OUT/write_$name
}
profile 25fv47 --carry 0
profile 25fv47_lui --carry 2
profile 25fv47_lui_device_refactor --carry 2 --lu-refactor 1
profile dense4096_f64 --workload dense4096 --dense-storage f64
cd $R
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default_stdout.txt 2> $OUT/bench_default.err   # the driver's invocation
tail -1 $OUT/bench_default_stdout.txt > $OUT/bench_default_line.json
cp bench_configs.json $OUT/bench_configs.json
ls -la $OUT
