#!/bin/bash
# Everything under profiles/<tag>_* in one GPU job (run from the repo root on an MI355X box):  bash tools/collect_round.sh r02
# rocprofv3 runs from /tmp; counter passes are separate from the kernel-trace / stats pass (MI355X_MICROARCH.md).
TAG=${1:-r02}
export TMPDIR=/tmp
R=$(pwd)
P=$R/gpurun_out/profiles
mkdir -p $P
timeout 900 python tools/collect_profiles.py $TAG > $R/gpurun_out/${TAG}_collect.log 2>&1
for W in train_step fed_step eval_pass seg_bwd kg_rank kg_pass; do
  rm -rf /tmp/kp_$W
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_$W -- python $R/tools/pmc_workloads.py $W > /dev/null 2>&1)
  F=$(find /tmp/kp_$W -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $P/${TAG}_${W}_kernel_stats.csv
done
for W in fed_step eval_pass seg_bwd; do
  : > $P/${TAG}_${W}_pmc.txt
  for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    rm -rf /tmp/pm_$W
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pm_$W -- python $R/tools/pmc_workloads.py $W > /dev/null 2>&1)
    F=$(find /tmp/pm_$W -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && python tools/pmc_summary.py $F eval_pass pspace_ topk_merge clip_step_kernel pref_bwd_wide_kernel kg_step_kernel feed_ pref_bwd_mc seg_reduce kg_bwd_rowout pref_fwd_mc >> $P/${TAG}_${W}_pmc.txt
  done
done
timeout 600 python tools/kernel_times.py > $P/${TAG}_kernel_times.txt 2>/dev/null
(timeout 300 python tools/config5_step.py; timeout 300 python tools/config5_step.py --zipf 1.05) > $P/${TAG}_config5_step.txt 2>/dev/null
timeout 900 python tools/cli_throughput.py > $P/${TAG}_cli_throughput.txt 2>/dev/null
timeout 300 python tools/kg_eval_pass.py > $P/${TAG}_kg_eval_pass.txt 2>/dev/null
timeout 900 python bench.py > $P/${TAG}_bench.json 2>/dev/null
ls -la $P
