#!/bin/bash
# builds examples/tracker_frame.cpp against the in-tree library and runs it: per-stage latencies of the tracker's per-frame chain
# usage: scripts/run_tracker_frame.sh [tag] [frames]   -> gpurun_out/tracker_<tag>.json (+ rocprofv3 kernel stats with PROF=1)
set -e
cd "$(dirname "$0")/.."
TAG=${1:-run}; FRAMES=${2:-300}
mkdir -p gpurun_out
EXE=/tmp/tracker_frame_$$
g++ -std=c++17 -O2 -o $EXE examples/tracker_frame.cpp -Lucoslam-cv3_amd -lucoslam_hip -Wl,-rpath,$PWD/ucoslam-cv3_amd -Wl,-rpath,/opt/rocm/lib -lpthread
$EXE 20 5 > /dev/null
$EXE $FRAMES 30 | tee gpurun_out/tracker_$TAG.json
if [ -n "$PROF" ]; then
  export TMPDIR=/tmp; OUT=$PWD/gpurun_out/tracker_prof_$TAG; rm -rf $OUT
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- $EXE 40 10 > /dev/null 2>&1) || true
  find $OUT -name "*kernel_stats.csv" -exec cp {} gpurun_out/tracker_kernel_stats_$TAG.csv \;
  find $OUT -name "*kernel_trace.csv" -exec cp {} gpurun_out/tracker_kernel_trace_$TAG.csv \;
  rm -rf $OUT
  head -30 gpurun_out/tracker_kernel_stats_$TAG.csv
fi
