R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_reg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --workload dense4096 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b.json 2> $OUT/stats.log
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/ks.csv \;
rm -rf $OUT/stats
cut -c80-130 $OUT/b.json
head -7 $OUT/ks.csv | cut -d, -f1-4 | cut -c1-150
