#!/bin/bash
# Round 4, runs on the GPU box (via gpurun).  Outputs under gpurun_out/prof_r04/, condensed by scripts/parse_profiles_r04.py into profiles/.
#   quick/    rocprofv3 --kernel-trace --stats of `bench.py --quick`: the headline loop ONLY (the dominant kernel's average is taken over
#             the same launches as ms_per_step)
#   pmc_*/    HBM traffic counters of the headline loop, one pass each (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc with --kernel-trace only)
#   tracker/  kernel stats of the tracker's per-frame chain (examples/tracker_frame.cpp, the C++ host)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r04
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="python $R/bench.py --quick --steps 20 --warmup 5 --reps 15"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/quick -o bench -- $Q > $OUT/quick.log 2>&1
Q3="python $R/bench.py --quick --steps 6 --warmup 2 --reps 1"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmc_fetch -o bench -- $Q3 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/pmc_write -o bench -- $Q3 > $OUT/pmc_write.log 2>&1
EXE=/tmp/tracker_frame_prof
g++ -std=c++17 -O2 -o $EXE $R/examples/tracker_frame.cpp -L$R/ucoslam-cv3_amd -lucoslam_hip -Wl,-rpath,$R/ucoslam-cv3_amd -Wl,-rpath,/opt/rocm/lib -lpthread
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/tracker -o trk -- $EXE 200 20 > $OUT/tracker.log 2>&1
$EXE 300 30 > $OUT/tracker_plain.json 2>&1
find $OUT -name "*.csv" | head -20
tail -2 $OUT/quick.log
