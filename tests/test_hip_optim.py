"""K20: fused global-norm clip + optimizer step vs torch.optim on CPU (the reference calls torch.optim directly,
utils/trainer.py:63-81, after clip_grad_norm over all parameters, e.g. item_recommendation.py:189-192)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)
SHAPES = [(6040, 100), (3240, 100), (20, 100), (7, 36), (1, 3)]      # incl. sizes that are not multiples of 4 / of a chunk
BIG = [(30000, 100), (7, 36), (1001, 3)]                             # more than ktup_optim_clip_step keeps in registers chip-wide


def make(kind, params, lr, wd, momentum):
    if kind == 'Adagrad':
        return torch.optim.Adagrad(params, lr=lr, weight_decay=wd)
    if kind == 'Adam':
        return torch.optim.Adam(params, lr=lr, weight_decay=wd)
    if kind == 'SGD':
        return torch.optim.SGD(params, lr=lr, weight_decay=wd, momentum=momentum)
    return torch.optim.RMSprop(params, lr=lr, weight_decay=wd, momentum=momentum)


@pytest.mark.parametrize('kind', ['Adagrad', 'Adam', 'SGD', 'Rmsprop'])
@pytest.mark.parametrize('wd', [0.0, 1e-5])
@pytest.mark.parametrize('max_norm', [0.05, 50.0])
@pytest.mark.parametrize('momentum', [0.0, 0.9])
@pytest.mark.parametrize('route', ['one_launch', 'two_launches', 'one_launch_big'])
def test_clip_and_step_matches_torch_optim(kind, wd, max_norm, momentum, route):
    """one_launch: ktup_optim_clip_step (norm, grid barrier, update on the registers of the norm pass); one_launch_big: the same
    kernel on tables too large for that (the update re-reads); two_launches: ktup_optim_gradnorm + ktup_optim_step."""
    if kind in ('Adagrad', 'Adam') and momentum != 0.0:
        pytest.skip('no momentum hyper-parameter')
    if route == 'one_launch_big' and (wd != 0.0 or momentum != 0.0):
        pytest.skip('covered at the small shapes')
    from jTransUP.utils.fused_optim import FusedOptimizer
    gen = torch.Generator().manual_seed(11)
    cpu = [torch.nn.Parameter(torch.randn(s, generator=gen) * 0.1) for s in (BIG if route == 'one_launch_big' else SHAPES)]
    gpu = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in cpu]
    ref = make(kind, cpu, 0.05, wd, momentum)
    fused = FusedOptimizer(make(kind, gpu, 0.05, wd, momentum))
    fused.one_launch = route != 'two_launches'
    moved = [0.0] * len(cpu)                               # cumulative size of the updates of each table
    for step in range(4):
        ref.zero_grad(set_to_none=False); fused.zero_grad()
        for i, (a, b) in enumerate(zip(cpu, gpu)):
            if step == 0 and i == 1:
                continue                                   # a table with no gradient yet is skipped by both
            if step == 2 and i == 0:
                g = torch.zeros_like(a)                    # touched before, zero gradient now: still decays
            else:
                g = torch.randn(a.shape, generator=gen) * (0.3 if step % 2 else 0.01)
            a.grad = g.clone(); b.grad = g.to(DEV)
        before = [a.detach().clone() for a in cpu]
        torch.nn.utils.clip_grad_norm_(cpu, max_norm)
        ref.step()
        fused.clip_and_step(max_norm)
        for i, (a, b, a0) in enumerate(zip(cpu, gpu, before)):
            # fp32 rounding of the updates (different fma contraction), measured against the size of the updates so far
            moved[i] += float((a.detach() - a0).abs().max())
            err = (b.detach().cpu() - a.detach()).abs()
            bad = err > 2e-4 * moved[i] + 2e-7
            # elements whose clipped gradient cancels the weight-decay term to ~eps are decided by the last bit of g + wd*p
            # (update = lr * d / (|d| + eps)): a handful per million, each bounded by the size of one update
            assert float(bad.float().mean()) <= 1e-4 and float(err.max()) <= 0.05 * moved[i] + 2e-7, \
                (kind, step, tuple(a.shape), float(err.max()), int(bad.sum()), moved[i])
            if a.grad is not None:
                # clipped in place like clip_grad_norm_; at 3M elements torch's own fp32 norm on the CPU is only good to ~1e-4
                torch.testing.assert_close(b.grad.cpu(), a.grad, rtol=2e-4 if route == 'one_launch_big' else 2e-5, atol=1e-7)
    assert fused.barrier_timeouts() == 0
    # the wrapped torch optimizer holds the state: its state_dict is interchangeable with the unfused run's
    sa, sb = ref.state_dict()['state'], fused.state_dict()['state']
    assert sa.keys() == sb.keys()
    for k in sa:
        assert sa[k].keys() == sb[k].keys()
        for name in sa[k]:
            if torch.is_tensor(sa[k][name]):
                x, y = sb[k][name].detach().cpu().float(), sa[k][name].float()
                bad = (x - y).abs() > 2e-4 * float(y.abs().max()) + 1e-7
                assert float(bad.float().mean()) <= 1e-4, (kind, k, name, int(bad.sum()))


def test_trainer_uses_the_fused_step(tmp_path):
    import logging
    from jTransUP.models import transUP
    from jTransUP.models.base import get_flags
    from jTransUP.utils.flags import FLAGS
    from jTransUP.utils.trainer import ModelTrainer
    get_flags(); FLAGS.reset()
    FLAGS(['prog', '-model_type', 'transup', '-log_path', str(tmp_path), '-experiment_name', 'opt'])
    FLAGS.ckpt_path = str(tmp_path)
    m = transUP.TransUPModel(False, 36, 30, 40, 5, False)
    tr = ModelTrainer(m, logging.getLogger('t'), 10, FLAGS)
    assert tr.fused is not None
    u = torch.randint(0, 30, (64,), device=DEV); i = torch.randint(0, 40, (64,), device=DEV)
    before = m.user_embeddings.weight.detach().clone()
    tr.optimizer_zero_grad()
    m(u, i).sum().backward()
    tr.clip_and_step(FLAGS.clipping_max_value)
    assert tr.step == 1 and not torch.equal(before, m.user_embeddings.weight.detach())


def test_adam_step_counts_host_array_and_device_array_agree():
    """ktup_optim_step takes Adam's step counts as a host array (`steps`) or from device memory (`steps_dev`, for graph
    replay): same update, and both equal torch.optim.Adam at that step."""
    import ctypes
    from jTransUP.hip import lib as L
    gen = torch.Generator().manual_seed(0)
    n, t = 5000, 7
    p0, g0 = torch.randn(n, generator=gen), torch.randn(n, generator=gen) * 0.1
    m0, v0 = torch.randn(n, generator=gen) * 0.01, torch.rand(n, generator=gen) * 0.01
    outs = []
    for use_dev in (False, True):
        p, g, m, v = (x.clone().to(DEV) for x in (p0, g0, m0, v0))
        arr = lambda vals: (ctypes.c_void_p * 1)(*vals)
        steps = (ctypes.c_int64 * 1)(t if not use_dev else 12345)            # ignored when steps_dev is given
        dsteps = torch.tensor([t], dtype=torch.int64, device=DEV)
        L.call('ktup_optim_step', 2, 1, arr([p.data_ptr()]), arr([g.data_ptr()]), arr([m.data_ptr()]), arr([v.data_ptr()]),
               (ctypes.c_int64 * 1)(n), steps, dsteps.data_ptr() if use_dev else None, (ctypes.c_int32 * 1)(0), 0.01, 0.0, 0.0, 0.9, 0.999,
               1e-8, 0.0, None, 0.0, 0, torch.cuda.current_stream().cuda_stream)
        outs.append((p.cpu(), m.cpu(), v.cpu()))
    for a, b in zip(*outs):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
    pt = torch.nn.Parameter(p0.clone()); pt.grad = g0.clone()
    opt = torch.optim.Adam([pt], lr=0.01)
    opt.state[pt] = {'step': torch.tensor(float(t - 1)), 'exp_avg': m0.clone(), 'exp_avg_sq': v0.clone()}
    opt.step()
    torch.testing.assert_close(outs[0][0], pt.data, rtol=1e-5, atol=1e-6)
