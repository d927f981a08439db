#!/bin/bash
# Round 3, runs on the GPU box (via gpurun).  Outputs under gpurun_out/prof_r03/, condensed by scripts/parse_profiles_r03.py into profiles/.
#   quick/   rocprofv3 --kernel-trace --stats of `bench.py --quick`: the headline loop ONLY (so that the dominant kernel's average is
#            taken over the same launches as ms_per_step)
#   stats/   the same of the full bench command (every side measurement included)
#   pmc_*/   HBM traffic counters of the headline loop, one pass each (FETCH_SIZE and WRITE_SIZE do not fit one pass)
#   shard/   kernel stats of the sharded frame stream with one rank (uh_fstream_*), through torch.distributed.run
#   mfma/    scripts/micro/mfma_f64_rate (issue rates of v_mfma_f64_16x16x4_f64 and v_fma_f64) and the MFMA / VALU busy counters of the
#            persistent BA kernel with its Schur product on MFMA (UH_BA_SCHUR=mfma) and on vector FMA
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT/mfma
cd /tmp && export TMPDIR=/tmp
Q="python $R/bench.py --quick --steps 20 --warmup 5 --reps 15"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/quick -o bench -- $Q > $OUT/quick.log 2>&1
F="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o bench -- $F > $OUT/stats.log 2>&1
Q3="python $R/bench.py --quick --steps 6 --warmup 2 --reps 1"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmc_fetch -o bench -- $Q3 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/pmc_write -o bench -- $Q3 > $OUT/pmc_write.log 2>&1
UH_BENCH_SHARDED=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/shard -o bench -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 1 --steps 5 --warmup 2 --reps 2 --no-cpu-baseline --no-roofline > $OUT/shard.log 2>&1
# ---- MFMA evidence
$R/scripts/micro/mfma_f64_rate.bin > $OUT/mfma/rate.txt 2>&1
rocprofv3 --list-avail 2>/dev/null | grep -i -E "mfma" | head -40 > $OUT/mfma/counters_available.txt
for form in mfma valu; do
  UH_BA_SCHUR=$form timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace -f csv -d $OUT/mfma/pmc_$form -o ba -- python $R/scripts/time_ba.py > $OUT/mfma/pmc_$form.log 2>&1
  UH_BA_SCHUR=$form timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --kernel-trace -f csv -d $OUT/mfma/pmc2_$form -o ba -- python $R/scripts/time_ba.py > $OUT/mfma/pmc2_$form.log 2>&1
  UH_BA_SCHUR=$form python $R/scripts/time_ba.py 2>/dev/null | head -1 > $OUT/mfma/time_$form.txt
done
find $OUT -name "*.csv" | head -40
