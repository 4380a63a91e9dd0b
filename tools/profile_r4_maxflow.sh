#!/bin/bash
# Round-4 profile of BASELINE config 5 from the reference's artificial start (`bench.py --workload maxflow --crash 0`): rocprofv3
# kernel stats, then the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, nothing beside --pmc).  Run on the GPU box from
# the repo root; the summaries land in gpurun_out/prof_r4_maxflow and are copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r4_maxflow
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--workload maxflow --crash 0 --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-concurrency-probe"
python3 $R/bench.py $ARGS > $OUT/bench_maxflow_reference_start.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_maxflow_reference_start.csv \;
rm -rf $OUT/stats
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/write.log
python3 $R/tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic_maxflow.json > $OUT/pmc_traffic_maxflow.txt 2>&1
rm -rf $OUT/fetch $OUT/write
ls -la $OUT
