import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, sys
dev = torch.device("cuda", 0)
W, i2e, idx = bench.build_world(3, dev)
D_ = {k: v.to(dev) for k, v in W.items()}
X = {k: v.to(dev) for k, v in idx.items()}
v = bench.variants_bench(dev, D_, i2e.to(dev, torch.int32), X)
print(' '.join('%s=%.4f' % (k[5:], x["ms_per_launch"]) for k, x in v.items() if k.startswith("eval")))
