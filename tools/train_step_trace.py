import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
print(bench.train_step_bench(torch.device('cuda', 0), steps=300)['ms_per_step_gpu_resident'])
