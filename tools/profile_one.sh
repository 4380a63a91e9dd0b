R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_greenbea
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RELP_GRAPH=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/solve_one.py GREENBEA > $OUT/out.txt 2> $OUT/stats.log
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/stats
head -8 $OUT/kernel_stats.csv | cut -c1-160
