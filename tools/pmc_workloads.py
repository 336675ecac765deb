"""Small fixed workloads for rocprofv3 counter passes (run under `rocprofv3 --kernel-trace --pmc ... -- python tools/pmc_workloads.py <name>`):
    eval_pass   the fused all-item evaluation sweep at ml1m shape (ktup_eval_pref_topk), 5 sweeps
    train_step  the three-launch B=512 joint training step, 20 rec + 20 kg steps
    fed_step    the device-fed B=512 joint step (-device_sampling): feed launch + step + clip/optimizer, ten-step graphs
    seg_bwd     the large-batch backwards by sorted segments: TransE (307,200 triples) and KTUP (716,800 pairs), 5 each
    kg_rank     the filtered gold ranks of one 512-query KG evaluation batch over 14,709 entities (ktup_eval_gold_ranks), 10 calls
    kg_pass     one direction of a link-prediction pass, 20,480 keys x 14,709 entities behind one call (ktup_eval_kg_ranks, TransH), 3 passes
    kg_pass_e   the same for TransE;  kg_pass_l1 / kg_pass_e_l1: with the L1 distance (the pair kernels' COUNT form)
    hard_pass   the ST-Gumbel gate's evaluation pass in one sweep (TUP, 6040 users x 3240 items, ktup_eval_pref_topk_hard), 3 passes,
                then the batched route over the same users (pairs_hard_kernel + K17 per 512 users), once
tools/pmc_summary.py turns the counter_collection.csv into per-kernel averages."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd'))
import torch

import bench as B


def eval_pass(dev):
    from jTransUP.models import jTransUP as jt
    torch.manual_seed(3)
    i_map = {i: i for i in range(B.NI)}
    new_map = {i: ((i * 4) % B.NE if i < B.ALIGNED else -1, i) for i in range(B.NI)}
    m = jt.jTransUPModel(False, B.D, B.NU, B.NI, B.NE, B.NR, i_map, new_map, False, False)
    m.eval(); m.disable_grad()
    u = torch.arange(B.NU, device=dev)
    gen = torch.Generator().manual_seed(1)
    f_off = (torch.arange(B.NU + 1) * 165).to(dev)
    f_ids = torch.randint(0, B.NI, (B.NU * 165,), generator=gen).to(dev, torch.int32)
    items = m.prepare_items()
    for _ in range(5):
        m.evaluate_topk(u, items, 10, f_off, f_ids)
    torch.cuda.synchronize()


def train_step(dev):
    B.train_step_bench(dev, steps=40, warmup=10)


def fed_step(dev):
    import types
    from jTransUP.models import jTransUP as jt
    from jTransUP.utils.device_sampler import DeviceSampler
    from jTransUP.utils.fast_train import DeviceFeeder, JointStepper
    from jTransUP.utils.fused_optim import FusedOptimizer
    torch.manual_seed(3)
    i_map = {i: i for i in range(B.NI)}
    new_map = {i: ((i * 4) % B.NE if i < B.ALIGNED else -1, i) for i in range(B.NI)}
    m = jt.jTransUPModel(False, B.D, B.NU, B.NI, B.NE, B.NR, i_map, new_map, False, False)
    opt = torch.optim.Adagrad(m.parameters(), lr=0.005, weight_decay=1e-5)
    tr = types.SimpleNamespace(fused=FusedOptimizer(opt), parameters=list(m.parameters()), model_target=-1, step=0)
    fl = types.SimpleNamespace(margin=1.0, kg_lambda=1.0, clipping_max_value=5.0)
    js = JointStepper(m, tr, fl, 512)
    gen = torch.Generator().manual_seed(6)
    ratings = torch.stack([torch.randint(0, B.NU, (96000,), generator=gen), torch.randint(0, B.NI, (96000,), generator=gen)], 1)
    triples = torch.stack([torch.randint(0, B.NE, (48000,), generator=gen), torch.randint(0, B.NE, (48000,), generator=gen),
                           torch.randint(0, B.NR, (48000,), generator=gen)], 1)
    sm = DeviceSampler(dev, seed=1)
    sm.set_rating_dicts(B.NU, B.NI, []); sm.set_triples(B.NE, B.NR, [triples.tolist()])
    js.attach_feeds(sm, rec=DeviceFeeder(ratings, 512, dev, seed=1), kg=DeviceFeeder(triples, 512, dev, seed=2))
    cyc = ('rec',) * 7 + ('kg',) * 3
    for s in range(20):
        js.fed_step(cyc[s % 10])
    for _ in range(10):
        if not js.fed_cycle(cyc):
            for k in cyc:
                js.fed_step(k)
    torch.cuda.synchronize()


def seg_bwd(dev):
    from jTransUP.hip import ops
    W, i2e, idx = B.build_world(3, dev)
    T = {k: v.to(dev) for k, v in W.items()}
    X = {k: v.to(dev) for k, v in idx.items()}
    i2e = i2e.to(dev, torch.int32)
    for _ in range(5):
        E, R = (T[k].detach().clone().requires_grad_(True) for k in ('E', 'R'))
        ops.score_transe(E, R, X['h'], X['t'], X['r'], False).sum().backward()
        tabs = [T[k].detach().clone().requires_grad_(True) for k in ('U', 'I', 'E', 'P', 'Pn', 'R', 'Rn')]
        ops.score_ktup(*tabs, i2e, X['u'], X['i'], False).sum().backward()
    torch.cuda.synchronize()


def kg_rank(dev):
    from jTransUP.hip import ops
    gen = torch.Generator().manual_seed(5)
    nq = 512
    scores = torch.rand(nq, B.NE, generator=gen).to(dev)
    g_off = (torch.arange(nq + 1) * 3).to(dev)
    g_ids = torch.randint(0, B.NE, (nq * 3,), generator=gen).to(dev, torch.int32)
    f_off = (torch.arange(nq + 1) * 40).to(dev)
    f_ids = torch.randint(0, B.NE, (nq * 40,), generator=gen).to(dev, torch.int32)
    for _ in range(10):
        ops.gold_ranks(scores, False, g_off, g_ids, f_off, f_ids)
    torch.cuda.synchronize()


def kg_pass(dev, transe=False, l1=False):
    from jTransUP.hip import ops
    gen = torch.Generator().manual_seed(7)
    nq = 20480
    E = torch.nn.functional.normalize(torch.randn(B.NE, B.D, generator=gen), dim=1).to(dev)
    R = torch.nn.functional.normalize(torch.randn(B.NR, B.D, generator=gen), dim=1).to(dev)
    N = torch.nn.functional.normalize(torch.randn(B.NR, B.D, generator=gen), dim=1).to(dev)
    q = torch.randint(0, B.NE, (nq,), generator=gen).to(dev); r = torch.randint(0, B.NR, (nq,), generator=gen).to(dev)
    # sorted sets per key, as the drivers' evaluation index hands them over
    strict = lambda n: (torch.sort(torch.randint(0, B.NE - n, (nq, n), generator=gen), dim=1)[0] + torch.arange(n)).reshape(-1)
    # 1-3 golds per key, as in the drivers' passes (and bench.py's eval_kg_bench): nearly every 64-key workgroup then holds a key with
    # three golds and runs the sweep's three-gold variant
    n_g = torch.randint(1, 4, (nq,), generator=gen)
    g_off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(n_g, 0)]).to(dev)
    keep = (torch.arange(3)[None, :] < n_g[:, None]).reshape(-1)
    g_ids = strict(3)[keep].to(dev, torch.int32)
    f_off = (torch.arange(nq + 1) * 20).to(dev)
    f_ids = strict(20).to(dev, torch.int32)
    for _ in range(3):
        ops.eval_kg_ranks(E, R, None if transe else N, q, r, l1, False, False, g_off, g_ids, f_off, f_ids)
    torch.cuda.synchronize()


def hard_pass(dev):
    from jTransUP.hip import ops
    from jTransUP.models import transUP as tu
    torch.manual_seed(3)
    m = tu.TransUPModel(False, B.D, B.NU, B.NI, B.NR, True).to(dev)
    m.eval(); m.disable_grad()
    u = torch.arange(B.NU, device=dev)
    gen = torch.Generator().manual_seed(1)
    f_off = (torch.arange(B.NU + 1) * 165).to(dev)
    f_ids = torch.randint(0, B.NI, (B.NU * 165,), generator=gen).to(dev, torch.int32)
    items = m.prepare_items()
    for _ in range(3):
        m.evaluate_topk(u, items, 10, f_off, f_ids)
    for s in range(0, B.NU, 512):
        ops.topk_filtered(m.evaluate(u[s:s + 512], items=items), False, 10, f_off[s:s + 513] - f_off[s], f_ids[int(f_off[s]):])
    torch.cuda.synchronize()


def soft_l1_pass(dev):
    """KTUP with -L1_flag (the reference's run scripts), soft gate: the whole pass in one sweep, then batch by batch."""
    from jTransUP.hip import ops
    from jTransUP.models import jTransUP as jt
    torch.manual_seed(3)
    i_map = {i: i for i in range(B.NI)}
    new_map = {i: ((i * 4) % B.NE if i < B.ALIGNED else -1, i) for i in range(B.NI)}
    m = jt.jTransUPModel(True, B.D, B.NU, B.NI, B.NE, B.NR, i_map, new_map, False, False).to(dev)
    m.eval(); m.disable_grad()
    u = torch.arange(B.NU, device=dev)
    gen = torch.Generator().manual_seed(1)
    f_off = (torch.arange(B.NU + 1) * 165).to(dev)
    f_ids = torch.randint(0, B.NI, (B.NU * 165,), generator=gen).to(dev, torch.int32)
    items = m.prepare_items()
    for _ in range(3):
        m.evaluate_topk(u, items, 10, f_off, f_ids)
    for s in range(0, B.NU, 512):
        ops.topk_filtered(m.evaluateRec(u[s:s + 512], items=items), False, 10, f_off[s:s + 513] - f_off[s], f_ids[int(f_off[s]):])
    torch.cuda.synchronize()


if __name__ == '__main__':
    {'soft_l1_pass': soft_l1_pass, 'kg_pass': kg_pass, 'kg_pass_e': lambda d: kg_pass(d, True), 'kg_pass_e_l1': lambda d: kg_pass(d, True, True), 'kg_pass_l1': lambda d: kg_pass(d, False, True), 'hard_pass': hard_pass, 'kg_rank': kg_rank, 'eval_pass': eval_pass, 'train_step': train_step, 'fed_step': fed_step, 'seg_bwd': seg_bwd}[sys.argv[1]](torch.device('cuda'))
