"""The B = 512 training step's figures of bench.py (train_step_bench: torch, fused optimizer, GPU-resident with and without the
tracked gradient norm, device-fed single steps and ten-step graphs) on their own, for A/B runs under an environment switch:

    KTUP_TRACKED_NORM=0 python tools/step_time.py        # the norm pass + grid barrier of rounds 2-3
    python tools/step_time.py [steps]

Prints one line of ms per step.  Needs a GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd'))
import torch

import bench as B

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
o = B.train_step_bench(torch.device('cuda'), steps=steps, warmup=50)
switches = {k: os.environ[k] for k in ('KTUP_TRACKED_NORM', 'KTUP_FEED_AHEAD', 'KTUP_FUSED_STEP') if k in os.environ}
print('STEP', switches, {k: round(v, 5) for k, v in o.items() if isinstance(v, float)})
