#!/bin/bash
# Round 6's final artefacts on the GPU box (via gpurun): GPU test summary, all-stage fuzz, kd / k-means builder timings, BA phase stamps,
# the bench line, then the rocprof passes (scripts/collect_profiles_r06.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_final
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" > $O/gpu_tests.txt
timeout 120 python scripts/ba_ab.py 10 3000 0 2 2>&1 | grep -v "^/opt" > $O/ba_ab.txt
UH_KD_CLK=1 timeout 120 python scripts/time_kdbuild.py 2>&1 | grep -v "^/opt" > $O/time_kdbuild.txt
UH_KM_TIMING=1 timeout 120 python scripts/time_hkmeans.py 2>&1 | grep -v "^/opt" | tail -12 > $O/time_hkmeans.txt
timeout 120 python scripts/time_orb.py 2>&1 | grep -v "^/opt" > $O/time_orb.txt
timeout 1800 python scripts/fuzz_parity.py 60 2>&1 | grep -v "^/opt" > $O/fuzz_all.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 1500 bash scripts/collect_profiles_r06.sh > $O/collect.log 2>&1
find $R/gpurun_out/prof_r06 -name "*kernel_trace.csv" -delete
find $R/gpurun_out/prof_r06 -name "*.csv" -size +20M -delete
cat $O/gpu_tests.txt; tail -3 $O/fuzz_all.txt; head -c 300 $O/bench.json
