#!/bin/bash
# Round-2 profiles of BASELINE config 5 (max-flow LP, V = 65 536, E = 1 048 576): rocprofv3 kernel stats and the two PMC passes
# (separate runs) of tools/maxflow_probe.py (plain launches: rocprofv3 does not survive hipGraph replay at this size), with the
# artificial start of the reference and with the crash basis; then the bench lines.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r2_maxflow
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RELP_IMPLICIT_BOUNDS=1 RELP_GRAPH=0
for crash in 1 0; do
  export RELP_CRASH=$crash
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_crash$crash -- python3 $R/tools/maxflow_probe.py 65536x1048576 > $OUT/probe_crash$crash.txt 2> $OUT/stats_crash$crash.log
  find $OUT/stats_crash$crash -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_crash$crash.csv \;
  rm -rf $OUT/stats_crash$crash
done
export RELP_CRASH=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/maxflow_probe.py 65536x1048576 > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/maxflow_probe.py 65536x1048576 > /dev/null 2> $OUT/write.log
python3 $R/tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic_crash1.json > $OUT/pmc_traffic_crash1.txt 2>&1
rm -rf $OUT/fetch $OUT/write
cd $R
unset RELP_GRAPH RELP_CRASH RELP_IMPLICIT_BOUNDS
python3 bench.py --workload maxflow --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_maxflow_crash.json 2> $OUT/bench_crash.log
python3 bench.py --workload maxflow --steps 1 --warmup 0 --crash 0 --no-cpu-baseline > $OUT/bench_maxflow_reference_start.json 2> $OUT/bench_nocrash.log
ls -la $OUT
