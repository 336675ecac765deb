#!/bin/bash
# The round's closing GPU job (one gpurun call, ~10 min): the whole GPU test suite, smoke, the bench line, and the profiles of the
# kernels changed last (pair-kernel routes).  bash tools/final_round.sh r03   -> gpurun_out/profiles/<tag>_*
TAG=${1:-r03}
export TMPDIR=/tmp
R=$(pwd)
P=$R/gpurun_out/profiles
mkdir -p $P
timeout 500 python -m pytest tests -m gpu -x -q > $R/gpurun_out/${TAG}_gpu_tests.txt 2>&1
grep -E "passed|failed|rror" $R/gpurun_out/${TAG}_gpu_tests.txt | tail -3
(python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -i "smoke\|error" | tail -3)
timeout 600 python bench.py > $P/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
for W in kg_pass_e_l1 kg_pass_l1 soft_l1_pass; do
  rm -rf /tmp/kp_$W
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_$W -- python $R/tools/pmc_workloads.py $W > /dev/null 2>&1)
  F=$(find /tmp/kp_$W -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $P/${TAG}_${W}_kernel_stats.csv
done
W=soft_l1_pass
: > $P/${TAG}_${W}_pmc.txt
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/pm_$W
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pm_$W -- python $R/tools/pmc_workloads.py $W > /dev/null 2>&1)
  F=$(find /tmp/pm_$W -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python tools/pmc_summary.py $F sweep_soft pairs_kernel >> $P/${TAG}_${W}_pmc.txt
done
: > $P/${TAG}_kg_eval_pass.txt
for V in "transh" "transe" "transe l1" "transh l1"; do timeout 200 python tools/kg_eval_pass.py $V >> $P/${TAG}_kg_eval_pass.txt 2>/dev/null; done
cat $P/${TAG}_kg_eval_pass.txt
