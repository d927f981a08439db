#!/bin/bash
# Round 5's final artefacts on the GPU box (via gpurun): GPU test summary, all-stage fuzz, BA fuzz walk, BA stress, window sweep, phase stamps,
# solve / chain-latency micro-benchmarks, the bench line, then the rocprof passes (scripts/collect_profiles_r05.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_final
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" > $O/gpu_tests.txt
python scripts/ba_ab.py 10 3000 0 2 2>&1 | grep -v "^/opt" > $O/ba_ab.txt
scripts/micro/solve48.bin > $O/solve48.txt 2>&1
scripts/micro/chain_latency.bin > $O/chain_latency.txt 2>&1
python scripts/fuzz_parity.py 60 2>&1 | grep -v "^/opt" > $O/fuzz_all.txt
python scripts/fuzz_parity.py 300 ba 2>&1 | grep -v "^/opt" > $O/fuzz_ba.txt
python scripts/ba_stress.py 60 2>&1 | grep -v "^/opt" | tail -5 > $O/ba_stress.txt
python scripts/ba_window_sweep.py 2>&1 | grep -v "^/opt" > $O/ba_window_sweep.txt
python scripts/time_pnp.py 2>&1 | grep -v "^/opt" > $O/time_pnp.txt
(UH_KNN_FORM=fused python scripts/knn_nq_sweep.py; python scripts/knn_nq_sweep.py) 2>&1 | grep -E "^fused|^default" > $O/knn_nq_sweep.txt
bash scripts/knn_pmc_r05.sh 2>&1 | grep -E "^8000|^2000" > $O/knn_pmc.txt
python bench.py > $O/bench.json 2> $O/bench.err
bash scripts/collect_profiles_r05.sh > $O/collect.log 2>&1
find $R/gpurun_out/prof_r05 -name "*kernel_trace.csv" -delete
find $R/gpurun_out/prof_r05 -name "*.csv" -size +20M -delete
cat $O/gpu_tests.txt; tail -2 $O/fuzz_all.txt; tail -2 $O/fuzz_ba.txt; cat $O/ba_stress.txt | tail -1; head -c 300 $O/bench.json
