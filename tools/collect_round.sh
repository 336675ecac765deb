#!/bin/bash
# Everything under profiles/<tag>_* in one GPU job (run from the repo root on an MI355X box):  bash tools/collect_round.sh r03
# rocprofv3 runs from /tmp; counter passes are separate from the kernel-trace / stats pass (MI355X_MICROARCH.md).
TAG=${1:-r06}
export TMPDIR=/tmp
R=$(pwd)
P=$R/gpurun_out/profiles
mkdir -p $P
timeout 900 python tools/collect_profiles.py $TAG > $R/gpurun_out/${TAG}_collect.log 2>&1
for W in train_step fed_step eval_pass seg_bwd kg_rank kg_pass kg_pass_e kg_pass_e_l1 hard_pass soft_l1_pass; do
  rm -rf /tmp/kp_$W
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_$W -- python $R/tools/pmc_workloads.py $W > /dev/null 2>&1)
  F=$(find /tmp/kp_$W -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $P/${TAG}_${W}_kernel_stats.csv
done
# config 5's step (the fixed-shape steppers, full-size tables): wall / device time of the rec step, the kg step and the 7 : 3 joint
# cycle in their forms, and the kernel-level profile of one rank (rec and kg, one-graph and exchange form)
for V in "--kind rec" "--kind kg" "--kind joint" "--kind rec --zipf 1.05" "--kind rec --no-overlap" "--kind rec --gradient-buffer --no-overlap" "--kind rec --exchange" "--kind joint --exchange"; do
  timeout 200 python tools/config5_step.py --steps 200 --full $V 2>/dev/null | grep config
done > $P/${TAG}_config5_step.txt
timeout 200 python tools/config5_step.py --steps 300 --full --kind joint --optimizer adam 2>/dev/null | grep config | sed 's/^/--optimizer adam /' >> $P/${TAG}_config5_step.txt
# Adam in its steady state (1,500 untimed steps first: every item and entity row then has a state and steps to replay whenever it is touched)
for V in "--kind joint" "--kind rec" "--kind joint --exchange"; do
  timeout 300 python tools/config5_step.py --steps 200 --full --optimizer adam --warm 1500 $V 2>/dev/null | grep config | sed "s/^/--optimizer adam --warm 1500 $V /"
done >> $P/${TAG}_config5_step.txt
timeout 200 python tools/config5_step.py --steps 200 --full --kind rec --exchange --segment-graphs 2>/dev/null | grep config | sed 's/^/--exchange --segment-graphs /' >> $P/${TAG}_config5_step.txt
# what runs beside what in one replayed step (start offsets, durations, hardware queues), one rank and exchange form
for V in "rec" "rec --exchange" "rec --optimizer adam --warm 1500"; do
  rm -rf /tmp/tl
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/tools/config5_step.py --steps 60 --full --kind $V > /dev/null 2>&1)
  echo "=== config5_step.py --full --kind $V"; python tools/trace_timeline.py /tmp/tl pref_bwd_wide 3
done > $P/${TAG}_config5_timeline.txt 2>&1
KTUP_WIDE_WAVES=4 timeout 200 python tools/config5_step.py --steps 200 --full --kind rec 2>/dev/null | grep config | sed 's/^/KTUP_WIDE_WAVES=4 /' >> $P/${TAG}_config5_step.txt
for V in rec kg rec_exchange rec_adam; do
  rm -rf /tmp/c5
  A="--kind rec"; [ $V = kg ] && A="--kind kg"; [ $V = rec_exchange ] && A="--kind rec --exchange"; [ $V = rec_adam ] && A="--kind rec --optimizer adam --warm 1500"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5 -- python $R/tools/config5_step.py --steps 100 --full --no-overlap $A > /dev/null 2>&1)
  F=$(find /tmp/c5 -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $P/${TAG}_config5_${V}_kernel_stats.csv
done
for W in fed_step eval_pass kg_pass hard_pass soft_l1_pass; do
  : > $P/${TAG}_${W}_pmc.txt
  for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    rm -rf /tmp/pm_$W
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pm_$W -- python $R/tools/pmc_workloads.py $W > /dev/null 2>&1)
    F=$(find /tmp/pm_$W -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && python tools/pmc_summary.py $F eval_pass pspace_ topk_merge clip_step_kernel pref_bwd_wide_kernel kg_step_kernel feed_ kg_count_mc kg_list_scores kg_rank_finalize kg_unc_resolve kg_wtab sweep_hard sweep_soft pairs_hard pairs_kernel >> $P/${TAG}_${W}_pmc.txt
  done
done
timeout 600 python tools/kernel_times.py > $P/${TAG}_kernel_times.txt 2>/dev/null
timeout 400 python tools/cli_throughput.py > $P/${TAG}_cli_throughput.txt 2>/dev/null
(timeout 200 python tools/step_profile.py fused; timeout 200 python tools/step_profile.py torch) 2>/dev/null | grep -A14 'STEP' | cut -c1-160 > $P/${TAG}_autograd_step_profile.txt
timeout 300 python tools/kg_eval_pass.py transh > $P/${TAG}_kg_eval_pass.txt 2>/dev/null
timeout 300 python tools/kg_eval_pass.py transe >> $P/${TAG}_kg_eval_pass.txt 2>/dev/null
timeout 300 python tools/kg_eval_pass.py transe l1 >> $P/${TAG}_kg_eval_pass.txt 2>/dev/null
timeout 300 python tools/kg_eval_pass.py transh l1 >> $P/${TAG}_kg_eval_pass.txt 2>/dev/null
(T=$(python -c 'import torch,os;print(os.path.dirname(torch.__file__))')/lib; echo "== LD_PRELOAD=torch/lib/libamdhip64.so"; LD_PRELOAD=$T/libamdhip64.so timeout 300 tools/graph_memset_repro; echo "== /opt/rocm/lib"; timeout 300 tools/graph_memset_repro) > $P/${TAG}_graph_memset_repro.txt 2>/dev/null
timeout 300 python tools/graph_memset_repro.py > $P/${TAG}_graph_memset_repro_torch.txt 2>/dev/null
timeout 120 tools/gumbel_log_check > $P/${TAG}_gumbel_log_check.txt 2>/dev/null
timeout 300 tools/gather_bench footprint > $P/${TAG}_gather_footprint.txt 2>/dev/null
# K6 forward at config 5's width on HBM-resident tables: the coordinate-split kernel and the one-wave-per-tile kernel it replaced, + its trace
(timeout 200 python tools/fwd_d256_time.py; KTUP_FWD_WIDE=0 timeout 200 python tools/fwd_d256_time.py) 2>/dev/null | grep FWD256 > $P/${TAG}_fwd_d256.txt
rm -rf /tmp/kp_fwd256
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_fwd256 -- python $R/tools/fwd_d256_time.py > /dev/null 2>&1)
F=$(find /tmp/kp_fwd256 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $P/${TAG}_fwd_d256_kernel_stats.csv
timeout 900 python bench.py > $P/${TAG}_bench.json 2>/dev/null
# the numbers a reader is shown come from the files themselves; the stamp ties them to the kernel sources they were measured on
python tools/profile_summary.py $TAG $P
ls -la $P
