#!/bin/bash
# Round-2 profiles: rocprofv3 kernel stats and the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs) of bench.py for both
# carries.  Run on the GPU box from the repo root; summaries land in gpurun_out/prof_r2 and are copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-dense-roofline --no-concurrency-probe"
for carry in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_carry$carry -- python3 $R/bench.py $ARGS --carry $carry > $OUT/bench_carry${carry}_under_rocprof.json 2> $OUT/stats_carry$carry.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_carry$carry -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dense-roofline --no-concurrency-probe --carry $carry > /dev/null 2> $OUT/fetch_carry$carry.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_carry$carry -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dense-roofline --no-concurrency-probe --carry $carry > /dev/null 2> $OUT/write_carry$carry.log
  python3 $R/tools/pmc_traffic.py $OUT/fetch_carry$carry $OUT/write_carry$carry $OUT/pmc_traffic_carry$carry.json > $OUT/pmc_traffic_carry$carry.txt 2>&1
  find $OUT/stats_carry$carry -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_carry$carry.csv \;
  rm -rf $OUT/fetch_carry$carry $OUT/write_carry$carry
  find $OUT/stats_carry$carry -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
cd $R && python3 bench.py --carry 0 > $OUT/bench_default.json 2> /dev/null
python3 bench.py --carry 1 --no-dense-roofline --no-cpu-baseline > $OUT/bench_lu.json 2> /dev/null
ls -la $OUT
