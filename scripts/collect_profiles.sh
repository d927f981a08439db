#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of the bench command + HBM traffic counters in separate passes
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Outputs land in gpurun_out/prof_${ROUND:-r02}/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_${ROUND:-r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1
CMD2="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmc_fetch -o bench -- $CMD2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/pmc_write -o bench -- $CMD2 > $OUT/pmc_write.log 2>&1
ls -R $OUT | head -30
