cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_hip_optim.py tests/test_fast_train.py tests/test_parallel_gloo.py tests/test_hip_config5.py tests/test_hip_segbwd.py tests/test_hip_score.py -x -q -m gpu --timeout 300 > $O/r2_t15.log 2>&1
tail -3 $O/r2_t15.log
timeout 300 python -c "import torch,bench,json; print(json.dumps({k:v for k,v in bench.train_step_bench(torch.device('cuda',0), steps=300).items() if 'ms' in k}))" > $O/r2_step15.json 2>$O/r2_step15.err
cat $O/r2_step15.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r2_prof_step15 -o s -- python tools/pmc_workloads.py train_step > $O/r2_prof_step15.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r2_prof_seg15 -o s -- python tools/pmc_workloads.py seg_bwd > $O/r2_prof_seg15.log 2>&1
