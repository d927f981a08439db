#!/bin/bash
# Round 6, runs on the GPU box (via gpurun).  Outputs under gpurun_out/prof_r06/, condensed by scripts/parse_profiles_r06.py into profiles/.
#   quick/      rocprofv3 --kernel-trace --stats of `bench.py --quick`: the headline loop ONLY (the dominant kernel's average over the same
#               launches as ms_per_step)
#   pmc_*/      HBM traffic counters of the headline loop, one pass each (--pmc with --kernel-trace only)
#   chain/      kernel stats of the local-BA launch chain at 17 / 24 / 32 / 48 / 64 free keyframes (scripts/ba_window_sweep.py): every MFMA
#               kernel of the path by name
#   chain_mfma/ SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES of the same command (its own pass), chain_fetch/, chain_write/: its traffic
#   tracker/    kernel stats of the tracker's per-frame chain (examples/tracker_frame.cpp, the C++ host)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r06
PARTS=${UH_COLLECT_PARTS:-headline tracker}   # e.g. UH_COLLECT_PARTS="headline tracker" refreshes those and leaves chain/ as it is
mkdir -p $OUT
has() { [[ " $PARTS " == *" $1 "* ]]; }
cd /tmp && export TMPDIR=/tmp
if has headline; then
rm -rf $OUT/quick $OUT/pmc_fetch $OUT/pmc_write
Q="python $R/bench.py --quick --steps 20 --warmup 5 --reps 15"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/quick -o bench -- $Q > $OUT/quick.log 2>&1
Q3="python $R/bench.py --quick --steps 6 --warmup 2 --reps 1"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmc_fetch -o bench -- $Q3 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/pmc_write -o bench -- $Q3 > $OUT/pmc_write.log 2>&1
fi
if has chain; then
rm -rf $OUT/chain $OUT/chain_mfma $OUT/chain_mops $OUT/chain_fetch $OUT/chain_write
export UH_SWEEP_NO_ORACLE=1
W="python $R/scripts/ba_window_sweep.py 3000 17 24 32 48 64"
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/chain -o chain -- $W > $OUT/chain.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -f csv -d $OUT/chain_mfma -o chain -- $W > $OUT/chain_mfma.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES --kernel-trace -f csv -d $OUT/chain_mops -o chain -- $W > $OUT/chain_mops.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/chain_fetch -o chain -- $W > $OUT/chain_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/chain_write -o chain -- $W > $OUT/chain_write.log 2>&1
fi
if has tracker; then
rm -rf $OUT/tracker
EXE=/tmp/tracker_frame_prof
g++ -std=c++17 -O2 -o $EXE $R/examples/tracker_frame.cpp -L$R/ucoslam-cv3_amd -lucoslam_hip -Wl,-rpath,$R/ucoslam-cv3_amd -Wl,-rpath,/opt/rocm/lib -lpthread
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/tracker -o trk -- $EXE 200 20 > $OUT/tracker.log 2>&1
$EXE 300 30 > $OUT/tracker_plain.json 2>&1
$EXE 300 30 dev > $OUT/tracker_dev_plain.json 2>&1
$EXE 300 30 fused > $OUT/tracker_fused_plain.json 2>&1
rm -rf $OUT/tracker_dev $OUT/hkmeans $OUT/tracker_fused
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/tracker_fused -o trk -- $EXE 200 20 fused > $OUT/tracker_fused.log 2>&1
python $R/scripts/trace_gaps.py $OUT/tracker_fused ingest > $OUT/tracker_fused_timeline.txt 2>&1
find $OUT/tracker_fused -name "*kernel_trace.csv" -delete
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/tracker_dev -o trk -- $EXE 200 20 dev > $OUT/tracker_dev.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/hkmeans -o hk -- python $R/scripts/time_hkmeans.py > $OUT/hkmeans.log 2>&1
fi
find $OUT -name "*.csv" | head -30
tail -2 $OUT/quick.log
