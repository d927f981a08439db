#!/bin/bash
# A short refresh of the round-5 artefacts that depend on the exact matcher / extractor kernels (the full set: scripts/collect_final_r05.sh):
# GPU test summary, the matcher's fuzz stages and sweeps, the bench line, kernel stats + traffic of the headline loop, the tracker chain.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_final
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" > $O/gpu_tests.txt
python scripts/fuzz_parity.py 45 orb,knn_exact,knn_twophase,knn_stream,kmeans_index,bow_matcher 2>&1 | grep -v "^/opt" > $O/fuzz_matcher.txt
(UH_KNN_FORM=fused python scripts/knn_nq_sweep.py; python scripts/knn_nq_sweep.py; UH_KNN_FORM=fused python scripts/knn_nq_sweep.py 2; python scripts/knn_nq_sweep.py 2) 2>&1 | grep -E "^fused|^default" > $O/knn_nq_sweep.txt
bash scripts/knn_pmc_r05.sh 2>&1 | grep -E "^8000|^2000" > $O/knn_pmc.txt
python scripts/ba_ab.py 10 3000 0 2 2>&1 | grep -v "^/opt" > $O/ba_ab.txt
scripts/micro/solve48.bin > $O/solve48.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
UH_COLLECT_PARTS="headline tracker" bash scripts/collect_profiles_r05.sh > $O/collect.log 2>&1
find $R/gpurun_out/prof_r05 -name "*kernel_trace.csv" -delete
find $R/gpurun_out/prof_r05 -name "*.csv" -size +20M -delete
cat $O/gpu_tests.txt; tail -2 $O/fuzz_matcher.txt; cat $O/knn_nq_sweep.txt; head -c 300 $O/bench.json
