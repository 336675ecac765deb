"""Global-norm clip + dense optimizer step as one HIP launch (ktup_optim_clip_step, K20; two launches -- ktup_optim_gradnorm /
ktup_optim_step -- with KTUP_CLIP_STEP=0).

The reference ends every training step with `clip_grad_norm(all params, clipping_max_value); optimizer.step()`
(item_recommendation.py:189-192, knowledge_representation.py:209-211, knowledgable_recommendation.py:399-401) on
torch.optim.{Adagrad, Adam, SGD, RMSprop} with weight_decay = l2_lambda (utils/trainer.py:63-77).  FusedOptimizer wraps the
very torch.optim object the trainer creates: hyper-parameters and state tensors stay in torch's own layout (so
`state_dict()` / `load_state_dict()` and checkpoints are interchangeable with an unfused run); only the arithmetic of
clip + step moves into the library."""
import ctypes
import os

import torch
import torch.optim as optim

from jTransUP.hip import lib as L

MAX_TENSORS = 12
OPTIM_WS_DOUBLES = 784                          # KTUP_OPTIM_WS_DOUBLES of include/ktup_hip.h
KINDS = {optim.SGD: 0, optim.Adagrad: 1, optim.Adam: 2, optim.RMSprop: 3}


def _arr(ctype, values):
    return (ctype * len(values))(*values)


class FusedOptimizer(object):
    def __init__(self, optimizer):
        if type(optimizer) not in KINDS:
            raise L.KtupError('FusedOptimizer supports SGD, Adagrad, Adam and RMSprop (got %s)' % type(optimizer).__name__)
        if len(optimizer.param_groups) != 1:
            raise L.KtupError('FusedOptimizer expects the single parameter group utils/trainer.py builds')
        self.optimizer = optimizer
        self.kind = KINDS[type(optimizer)]
        g = optimizer.param_groups[0]
        if g.get('maximize') or g.get('amsgrad') or g.get('nesterov') or g.get('centered') or g.get('dampening', 0) != 0 \
                or g.get('lr_decay', 0) != 0:
            raise L.KtupError('FusedOptimizer implements the reference configuration only (no maximize / amsgrad / nesterov / '
                              'centered / dampening / lr_decay)')
        self._sumsq = None
        self.one_launch = os.environ.get('KTUP_CLIP_STEP', '1') != '0'
        self._plan = None
        self._fresh = []
        self._dev_steps = None                   # Adam: step counts in device memory (the launch is then graph-replayable)
        self._dev_steps_for = None
        self._replayed = 0                       # steps replayed from a graph whose host-side counters are not written back yet

    # ---- torch.optim surface the trainer uses
    def zero_grad(self):
        """Zero-FILL, like torch 0.3's zero_grad: a table that has received a gradient once keeps being updated (weight
        decay, moment decay) on steps that do not touch it, e.g. user/item tables on KTUP's KG steps."""
        self.optimizer.zero_grad(set_to_none=False)

    def state_dict(self):
        self._flush_steps()
        return self.optimizer.state_dict()

    def load_state_dict(self, sd):
        self._replayed = 0
        self.optimizer.load_state_dict(sd)
        self._plan = None                        # the state tensors were replaced
        self._dev_steps_for = None

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    # ---- state in torch's layout (created the way each torch optimizer creates it)
    def _state(self, p, group):
        """Create the state tensors the way each torch optimizer does; returns (state1, state2, momentum buffer is fresh)."""
        if self.kind == 0 and group['momentum'] == 0:
            return None, None, 0                 # plain SGD keeps no state (torch leaves optimizer.state empty)
        st = self.optimizer.state[p]
        first = 0
        if self.kind == 0:                       # SGD: momentum_buffer (None until the first step)
            if st.get('momentum_buffer') is None:
                st['momentum_buffer'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                first = 1
            return st.get('momentum_buffer'), None, first
        if 'step' not in st:
            st['step'] = torch.tensor(0.0, dtype=torch.float32)
        if self.kind == 1:                       # Adagrad: 'sum' exists from construction
            if 'sum' not in st:
                st['sum'] = torch.full_like(p, group.get('initial_accumulator_value', 0.0), memory_format=torch.preserve_format)
            s1, s2 = st['sum'], None
        elif self.kind == 2:
            if 'exp_avg' not in st:
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            s1, s2 = st['exp_avg'], st['exp_avg_sq']
        else:
            if 'square_avg' not in st:
                st['square_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if group['momentum'] > 0:
                    st['momentum_buffer'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            s1, s2 = st['square_avg'], st.get('momentum_buffer')
        return s1, s2, first

    def sumsq_ptr(self, dev):
        """Device workspace of the clip: [0] = the squared gradient norm of the last clipped step, the rest is the one-launch
        kernel's barrier scratch (zero-filled once, here)."""
        if self._sumsq is None or self._sumsq.device != dev:
            self._sumsq = torch.zeros(OPTIM_WS_DOUBLES, dtype=torch.float64, device=dev)
        return self._sumsq.data_ptr()

    def barrier_timeouts(self):
        """Non-zero if a ktup_optim_clip_step launch ever gave up waiting at its grid barrier (device -> host sync; tests)."""
        return 0 if self._sumsq is None else int(self._sumsq[OPTIM_WS_DOUBLES - 1:].view(torch.int64).item())

    @torch.no_grad()
    def clip_and_step(self, max_norm, zero_grads=False, loss=None, gnorm=None):
        """`zero_grads`: leave the gradients zero-filled (the next step's zero_grad folded into this pass) instead of
        clipped in place.  `loss` = (slots_ptr, n_slots, scale, out_ptr[, acc_ptr]): the fused training step's loss slots are folded
        into *out (and added to the running sum *acc) and cleared by the same launch (needs the one-launch route).  `gnorm`: pointer
        to the gradient-norm workspace the step launch tracked the squared norm in (include/ktup_hip.h KTUP_GNORM_WS_DOUBLES): the
        launch then has no norm pass and no grid barrier."""
        self._flush_steps()
        group = self.optimizer.param_groups[0]
        ps = [p for p in group['params'] if p.grad is not None]
        if not ps:
            return
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps) + (torch.cuda.current_stream(ps[0].device).cuda_stream,)
        plan = self._plan if self._plan is not None and self._plan[0] == key else self._make_plan(ps, key, group)
        _, n, params, grads, s1, s2, sizes, stream, dev = plan
        steps, firsts = [], []
        for p in ps:                                            # per-tensor step counts (Adam bias correction), first-use flags
            st = self.optimizer.state.get(p)
            if st is not None and 'step' in st:
                st['step'] += 1
                steps.append(int(st['step'].item()))
            else:
                steps.append(1)
            firsts.append(0)
        if self.kind == 0 and group['momentum'] != 0:
            firsts = [1 if f else 0 for f in self._fresh]
            self._fresh = [False] * n
        steps_dev = None
        if self.kind == 2:                                      # device copy of the counters, advanced by a (capturable) launch
            if self._dev_steps_for != key[:-1]:
                self._dev_steps = torch.tensor([s - 1 for s in steps], dtype=torch.int64, device=dev)
                self._dev_steps_for = key[:-1]
            self._dev_steps.add_(1)
            steps_dev = self._dev_steps.data_ptr()
        clip = max_norm is not None and max_norm > 0
        betas = group.get('betas', (0.9, 0.999))
        hyper = (float(group['lr']), float(group['weight_decay']), float(group.get('momentum', 0.0)), float(betas[0]), float(betas[1]),
                 float(group.get('eps', 0.0)), float(group.get('alpha', 0.0)))
        head = (self.kind, n, params, grads, s1, s2, sizes, _arr(ctypes.c_int64, steps), steps_dev, _arr(ctypes.c_int32, firsts)) + hyper
        if gnorm is not None and not clip:
            gnorm = None
        if self.one_launch or loss is not None or gnorm is not None:
            lo = (None, 0, 0.0, None, None) if loss is None else (loss[0], int(loss[1]), float(loss[2]), loss[3], loss[4] if len(loss) > 4 else None)
            L.call('ktup_optim_clip_step', *head, self.sumsq_ptr(dev), gnorm, float(max_norm) if clip else 0.0, int(bool(zero_grads)), *lo, stream)
            return
        sumsq = None
        if clip:
            sumsq = self.sumsq_ptr(dev)
            L.call('ktup_optim_gradnorm', n, grads, sizes, sumsq, stream)
        L.call('ktup_optim_step', *head, sumsq, float(max_norm) if clip else 0.0, int(bool(zero_grads)), stream)

    def _make_plan(self, ps, key, group):
        """Validate once per set of (parameter, gradient) buffers and keep the pointer arrays: rebuilding them every step
        costs more host time than the two launches take on the device."""
        if len(ps) > MAX_TENSORS:
            raise L.KtupError('FusedOptimizer handles up to %d tables per step (got %d)' % (MAX_TENSORS, len(ps)))
        dev = ps[0].device
        for p in ps:
            if not p.is_cuda or p.device != dev or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() \
                    or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                raise L.KtupError('FusedOptimizer needs dense contiguous fp32 parameters and gradients on one MI355X device')
        s1l, s2l, fresh = [], [], []
        for p in ps:
            s1, s2, first = self._state(p, group)
            s1l.append(None if s1 is None else s1.data_ptr()); s2l.append(None if s2 is None else s2.data_ptr()); fresh.append(bool(first))
        self._fresh = fresh
        self._plan = (key, len(ps), _arr(ctypes.c_void_p, [p.data_ptr() for p in ps]), _arr(ctypes.c_void_p, [p.grad.data_ptr() for p in ps]),
                      _arr(ctypes.c_void_p, s1l), _arr(ctypes.c_void_p, s2l), _arr(ctypes.c_int64, [p.numel() for p in ps]),
                      torch.cuda.current_stream(dev).cuda_stream, dev)
        return self._plan

    def graph_safe(self):
        """True when a step has no host-computed, step-dependent launch arguments and can therefore be replayed from a
        captured HIP graph: always, now that Adam's step counts are also kept in device memory (bias corrections are
        evaluated by the kernel)."""
        return True

    def bump_steps(self, n=1):
        """Account for n steps whose launches were replayed from a graph (this object's Python code did not run).  The
        torch-side `state[p]['step']` tensors are brought up to date lazily -- before the next eager step and before
        state_dict() -- because touching seven CPU tensors per replay costs more host time than the replay itself."""
        self._replayed += int(n)

    def _flush_steps(self):
        if not self._replayed:
            return
        n, self._replayed = self._replayed, 0
        for p in self.optimizer.param_groups[0]['params']:
            st = self.optimizer.state.get(p)
            if p.grad is not None and st is not None and 'step' in st:
                st['step'] += n

    def total_norm(self):
        """Gradient norm of the last clipped step (device -> host sync; diagnostics only)."""
        return None if self._sumsq is None else float(self._sumsq[0].sqrt().item())
