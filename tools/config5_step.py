"""BASELINE config 5 (KTUP, d = 256, row-sharded user / item / entity tables): the per-GPU training step, timed.

    python tools/config5_step.py                       # one GPU: this rank owns every row (no exchange)
    torchrun --nproc-per-node 8 tools/config5_step.py  # rows partitioned by row % 8, RCCL all-to-all over xGMI

Per rank and step: B (u, pos, neg) triples.  lookup (unique ids -> [all-to-all -> pack -> all-to-all] -> compact rows),
KTUP scores of [pos ; neg] on the compact tables (K6, d = 256 matrix-core kernel), BPR loss, backward (dense gradients of
the compact rows only), ShardedStep.apply (row gradients back to their owners, global-norm clip, row-sparse Adagrad).
Table sizes are 1/8 of config 5 per rank (10 M users, 1 M items, 5 M entities over 8 GPUs) unless --full is given."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'joint-kg-recommender_amd'))

import torch
import torch.distributed as dist


def fused_main(args):
    """jTransUP/sharded_ktup.py: the step as HIP-graph replays, no host sync.  Reports wall time per step (host clock around
    `steps` replays, one synchronize at the end) and the device time between HIP events around the same replays."""
    from jTransUP import parallel
    from jTransUP.sharded_ktup import ShardedKgStepper, ShardedKtupJoint, ShardedKtupStepper
    rank, world = parallel.init_distributed()
    dev = torch.device('cuda', torch.cuda.current_device())
    scale = 1 if args.full else 8
    NU, NI, NE, P, d, B = 10_000_000 // scale * world, 1_000_000 // scale * world, 5_000_000 // scale * world, 20, args.d, args.batch
    gen = torch.Generator(device=dev); gen.manual_seed(3 + rank)

    def table(n):
        t = parallel.ShardedTable(n, d, rank=rank, world=world, device=dev)
        t.weight.data.normal_(generator=gen)
        t.weight.data.mul_(1.0 / 16.0)
        return t
    Ut, It, Et = table(NU), table(NI), table(NE)
    small = [torch.nn.Parameter(torch.nn.functional.normalize(torch.randn(P, d, generator=gen, device=dev), dim=1)) for _ in range(4)]
    if world > 1:
        for p in small:
            dist.broadcast(p.data, src=0)
    item2ent = torch.randint(0, NE, (NI,), generator=torch.Generator(device=dev).manual_seed(7), device=dev).to(torch.int32)
    st = ShardedKtupStepper(Ut, It, Et, *small, item2ent, batch=B, kind=args.optimizer, lr=0.005, max_norm=5.0, use_graphs=not args.no_graphs,
                            force_exchange=args.exchange, overlap_route=not args.no_overlap, fused_apply=not args.gradient_buffer, direct=False if (args.no_direct or args.exchange or world > 1) else None,
                            orth=args.kind != 'rec', route_beside=args.route_beside, exchange_graph=not args.segment_graphs)
    rec = st
    kg = None
    if args.kind != 'rec':             # the kg half of the joint schedule (knowledgable_recommendation.py:345-383) on the same entity shard
        kg = ShardedKgStepper(Et, small[2], small[3], batch=B, kind=args.optimizer, lr=0.005, max_norm=5.0, margin=1.0, kg_lambda=1.0,
                              small_state=rec.small_state[2:4], opt_step=rec.opt_step, use_graphs=not args.no_graphs, force_exchange=args.exchange,
                              overlap_route=not args.no_overlap, direct=False if (args.no_direct or args.exchange or world > 1) else None, exchange_graph=not args.segment_graphs)
        st = kg if args.kind == 'kg' else ShardedKtupJoint(rec, kg, 0.7)

    def draw(n_rows):
        if args.zipf <= 0:
            return torch.randint(0, n_rows, (B,), generator=gen, device=dev)
        uu = torch.rand(B, generator=gen, device=dev, dtype=torch.float64)
        a1 = args.zipf - 1.0
        top = float(n_rows) ** (-a1)
        r = (1.0 - uu * (1.0 - top)) ** (-1.0 / a1)
        return (r.clamp(1, n_rows) - 1).to(torch.int64)
    n = args.steps + args.warm
    n_draw = n + 2 * max(args.cycle, 0)
    batches = [(draw(NU), draw(NI), draw(NI)) for _ in range(n_draw)]
    if kg is not None:
        kb = []
        for _ in range(n_draw):
            ph, pt, other = draw(NE), draw(NE), draw(NE)
            pr = torch.randint(0, P, (B,), generator=gen, device=dev)
            flip = torch.rand(B, generator=gen, device=dev) < 0.5
            kb.append((ph, pt, pr, torch.where(flip, other, ph), torch.where(flip, pt, other), pr))
        kg.set_feed([torch.stack([b[c] for b in kb]).contiguous() for c in range(6)])
    if not args.copy_batches or kg is not None:              # device-fed: the step's own launches walk the pre-drawn columns
        rec.set_feed([torch.stack([b[c] for b in batches]).contiguous() for c in range(3)])
        batches = [()] * n_draw
    if kg is not None:
        run = st.run
    else:
        run = None
    for s in range(args.warm):
        run() if run else st(*batches[s])
    torch.cuda.synchronize(dev)
    l0 = float(rec.loss_sum[0])
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    if args.cycle > 0:
        assert args.steps % args.cycle == 0 and not args.copy_batches
        st.run_cycle(args.cycle); st.run_cycle(args.cycle)          # (capture + one replay, untimed: 2 x cycle more batches are drawn below)
        torch.cuda.synchronize(dev)
        l0 = float(rec.loss_sum[0])
        t0 = time.perf_counter()
        ev0.record()
        for s in range(args.warm, n, args.cycle):
            st.run_cycle(args.cycle)
    else:
        for s in range(args.warm, n):
            run() if run else st(*batches[s])
    ev1.record()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    devms = ev0.elapsed_time(ev1) / args.steps
    if world > 1:
        tt = torch.tensor([wall], device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); wall = float(tt)
    st.check()
    if rank == 0:
        print(json.dumps({'config': 'KTUP d=%d, %d/%d/%d rows (users/items/entities) over %d rank(s), B=%d per rank, ids %s' % (d, NU, NI, NE, world, B, 'Zipf(%.2f)' % args.zipf if args.zipf > 0 else 'uniform'),
                          'kind': args.kind,
                          'route': 'sharded_ktup.ShardedKtupStepper (%s%s%s%s)' % ('eager launches' if args.no_graphs else 'graph replay', ', exchange form' if args.exchange else '',
                                                                                    ', direct gathers' if rec.direct else ', packed rows', ', batches copied in' if args.copy_batches else ', device-fed'),
                          'graph_steps': args.cycle if args.cycle > 0 else 1, 'ms_per_step': 1e3 * wall / args.steps, 'ms_per_step_device': devms,
                          'scored_rows_per_s': 2 * B * world * args.steps / wall, 'wire_rows': (kg if args.kind == 'kg' else rec).W,
                          'mean_loss': (float(rec.loss_sum[0]) - l0) / args.steps}))
    rec.close()
    if kg is not None:
        kg.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8192)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warm', type=int, default=10, help='untimed steps before the timed ones (Adam / weight decay: a few hundred bring the item rows to their steady state -- every touched row then has steps to replay)')
    ap.add_argument('--d', type=int, default=256)
    ap.add_argument('--optimizer', default='adagrad', choices=['adagrad', 'sgd', 'adam'], help='adam: the row-sparse Adam with catch-up of the untouched steps (ktup_adam_t)')
    ap.add_argument('--kind', default='rec', choices=['rec', 'kg', 'joint'], help='rec: the rec step alone (the round-3 figure); kg: the kg step alone; joint: the 7 : 3 cycle of knowledgable_recommendation.py:320')
    ap.add_argument('--route-beside', action='store_true', help='rec step: the step kernel reads the id columns itself, the WHOLE route (init launch included) runs on the second graph branch')
    ap.add_argument('--zipf', type=float, default=0.0, help='draw ids from Zipf(a) (hot rows: contention in the row-gradient atomics) instead of uniformly, e.g. 1.05')
    ap.add_argument('--full', action='store_true', help='the whole 10M / 1M / 5M tables on this rank set (needs ~17 GB per rank at world 1)')
    ap.add_argument('--legacy', action='store_true', help="round 2's route: parallel.ShardedStep through autograd (eager torch ops around the kernels)")
    ap.add_argument('--exchange', action='store_true', help='one rank in exchange form: the several-ranks route (five segments, all-to-alls) talking to itself')
    ap.add_argument('--no-graphs', action='store_true')
    ap.add_argument('--segment-graphs', action='store_true', help='exchange form: one graph per segment with the collectives issued between the replays (what gloo runs) instead of the whole step as one graph')
    ap.add_argument('--copy-batches', action='store_true', help='hand every batch over as three tensors (three device copies per step) instead of device-fed columns')
    ap.add_argument('--no-overlap', action='store_true', help='one rank, direct gathers: keep the route on the step kernel\'s stream (no second stream in the graph)')
    ap.add_argument('--gradient-buffer', action='store_true', help='reduce -> norm -> apply through a W x d gradient buffer (three launches) instead of two walks over the per-pair gradients')
    ap.add_argument('--cycle', type=int, default=0, help='one rank, device-fed: the timed steps in graphs of this many steps each (run_cycle; even, a divisor of --steps)')
    ap.add_argument('--no-direct', action='store_true', help='one rank: pack the rows into the compact wire table first (what several ranks do)')
    args = ap.parse_args()
    if not args.legacy:
        return fused_main(args)
    from jTransUP import parallel
    from jTransUP.hip import ops
    from jTransUP.utils import loss as Lf
    rank, world = parallel.init_distributed()
    dev = torch.device('cuda', torch.cuda.current_device())
    scale = 1 if args.full else 8
    NU, NI, NE, P, d, B = 10_000_000 // scale * world, 1_000_000 // scale * world, 5_000_000 // scale * world, 20, args.d, args.batch
    gen = torch.Generator(device=dev); gen.manual_seed(3 + rank)
    def table(n):
        t = parallel.ShardedTable(n, d, rank=rank, world=world, device=dev)
        t.weight.data.copy_(torch.nn.functional.normalize(torch.randn(t.weight.shape, generator=gen, device=dev), dim=1))
        return t
    Ut, It, Et = table(NU), table(NI), table(NE)
    small = [torch.nn.Parameter(torch.nn.functional.normalize(torch.randn(P, d, generator=gen, device=dev), dim=1)) for _ in range(4)]
    Pm, Pn, R, Rn = small
    item2ent = torch.randint(0, NE, (NI,), generator=gen, device=dev)          # alignment map (replicated, int64 here)
    step = parallel.ShardedStep('adagrad', lr=0.005, max_norm=5.0)
    times = {'lookup': 0.0, 'score_fwd_bwd': 0.0, 'apply': 0.0}
    def tick():
        torch.cuda.synchronize(dev)
        return time.perf_counter()
    losses = []
    def draw(n_rows):
        if args.zipf <= 0:
            return torch.randint(0, n_rows, (B,), generator=gen, device=dev)
        # Zipf(a) by inverse transform of the continuous power law on [1, n_rows]: rank ~ u^(-1/(a-1)) truncated (a > 1)
        uu = torch.rand(B, generator=gen, device=dev, dtype=torch.float64)
        a1 = args.zipf - 1.0
        top = float(n_rows) ** (-a1)
        rank = (1.0 - uu * (1.0 - top)) ** (-1.0 / a1)
        return (rank.clamp(1, n_rows) - 1).to(torch.int64)
    for s in range(args.steps + 3):
        u = draw(NU)
        pi = draw(NI); ni = draw(NI)
        t0 = tick()
        items = torch.cat([pi, ni])
        # all three tables through one id exchange and one row exchange (no exchange and no host sync at all on one rank);
        # every item's entity row travels too (no pad rows here)
        (u_rows, u_at), (i_rows, i_at), (e_rows, e_at) = step.lookup_many([(Ut, u), (It, items), (Et, item2ent[items])])
        t1 = tick()
        # the scorer addresses the compact tables: item k of the batch -> compact item row i_at[k], compact entity row e_at[k];
        # item2ent for the compact item table = the entity position of (one of) the batch entries that produced that row
        i2e_compact = torch.zeros(i_rows.shape[0], dtype=torch.int32, device=dev)
        i2e_compact[i_at] = e_at.to(torch.int32)
        uu = torch.cat([u_at, u_at])
        score = ops.score_ktup(u_rows, i_rows, e_rows, Pm, Pn, R, Rn, i2e_compact, uu, i_at, False, ent_pad=-1)
        loss = Lf.bprLoss(score[:B], score[B:], target=-1) / world              # global mean over world * B
        loss.backward()
        t2 = tick()
        step.apply(replicated=small)
        t3 = tick()
        if s >= 3:
            times['lookup'] += t1 - t0; times['score_fwd_bwd'] += t2 - t1; times['apply'] += t3 - t2
            losses.append(float(loss.detach()) * world)
    total = sum(times.values())
    if world > 1:
        tt = torch.tensor([total], device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); total = float(tt)
    if rank == 0:
        print(json.dumps({'config': 'KTUP d=%d, %d/%d/%d rows (users/items/entities) over %d rank(s), B=%d per rank, ids %s' % (d, NU, NI, NE, world, B, 'Zipf(%.2f)' % args.zipf if args.zipf > 0 else 'uniform'),
                          'ms_per_step': 1e3 * total / args.steps, 'ms': {k: 1e3 * v / args.steps for k, v in times.items()},
                          'scored_rows_per_s': 2 * B * world * args.steps / total, 'loss_first_last': [losses[0], losses[-1]]}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
