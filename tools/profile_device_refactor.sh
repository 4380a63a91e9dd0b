#!/bin/bash
# rocprofv3 kernel stats of the 25FV47 / GREENBEA solves under RELP_CARRY_LU_INVERSE with the refactorisation on the device (lu_refactor = 1):
# where a device refactorisation's time goes, kernel by kernel.  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_refactor
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RELP_PROBE_CARRY=2 RELP_REFACTOR=device
for name in 25FV47 GREENBEA; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 $R/tools/refactor_probe.py $name > $OUT/${name}.txt 2> $OUT/stats_$name.log
  find $OUT/stats_$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_device_refactor_$name.csv \;
  rm -rf $OUT/stats_$name
done
RELP_TIME_REFACTOR=1 python3 $R/tools/refactor_probe.py 25FV47 GREENBEA > $OUT/timed.txt 2>&1
head -12 $OUT/kernel_stats_device_refactor_25FV47.csv $OUT/kernel_stats_device_refactor_GREENBEA.csv | cut -c1-200
tail -20 $OUT/timed.txt
