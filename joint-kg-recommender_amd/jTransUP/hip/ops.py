"""torch-facing wrappers of the HIP scoring kernels: argument checking, output allocation, autograd.

PyTorch is plumbing here (device memory, streams, autograd graph); all arithmetic on the hot path is
done by libktup_hip.so through its C ABI.  CPU tensors are rejected: there is no fallback path.
"""
import torch
from torch.autograd import Function

from . import lib as L

GUMBEL_OFF, GUMBEL_INPUT, GUMBEL_PHILOX, GUMBEL_PHILOX_DEV = 0, 1, 2, 3


def _dev(t):
    if not t.is_cuda:
        raise L.KtupError('ktup HIP ops need tensors on an MI355X device (got a CPU tensor); '
                          'there is no CPU fallback by design')
    return t.device


def _table(name, t):
    _dev(t)
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise L.KtupError('%s must be a 2-D fp32 table with unit inner stride (got %s %s strides %s)'
                          % (name, t.dtype, tuple(t.shape), t.stride()))
    return t


def _ids(name, t, dev, n=None):
    if t.device != dev:
        raise L.KtupError('%s must live on %s (got %s)' % (name, dev, t.device))
    if t.dtype != torch.int64 or t.dim() != 1:
        raise L.KtupError('%s must be a 1-D int64 index tensor (got %s %s)' % (name, t.dtype, tuple(t.shape)))
    if n is not None and t.numel() != n:
        raise L.KtupError('%s has %d entries, expected %d' % (name, t.numel(), n))
    return t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


def _seg_ws(nbytes, dev):
    """Scratch of the large-batch backward route (per-row gradients + the segment sort); None = the atomics route applies."""
    return torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=dev) if nbytes else None


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _zeros_like_many(*tables):
    """Zero-filled gradient buffers for several tables out of ONE allocation and ONE fill launch (a backward through autograd is bound
    by the host: ~8 us per torch call); every part starts on a 16-byte boundary."""
    sizes = [(t.numel() + 3) & ~3 for t in tables]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=tables[0].device)
    out, at = [], 0
    for t, sz in zip(tables, sizes):
        out.append(flat[at:at + t.numel()].view(t.shape))
        at += sz
    return out


_DIRECT_GRAD = [True]


def set_direct_grad(on):
    """Switch the gradient-accumulation fusion of the score Functions (see _grad_targets) on or off; returns the previous setting.
    Off is needed only by code that calls torch.autograd.grad(...) on tables whose `.grad` is populated."""
    was = _DIRECT_GRAD[0]
    _DIRECT_GRAD[0] = bool(on)
    return was


def _takes_grad_directly(t):
    if not (t.is_leaf and t.requires_grad):          # (.grad of a non-leaf warns)
        return False
    g = t.grad
    return (g is not None and g.dtype == torch.float32 and g.layout == torch.strided
            and g.device == t.device and g.shape == t.shape and g.is_contiguous() and t.dim() == 2 and t.stride(0) == t.shape[1]
            and not t._backward_hooks and not getattr(t, '_post_accumulate_grad_hooks', None))


def _grad_targets(*tables):
    """Where a backward's kernels add the table gradients -> (buffers for the kernels, values for autograd).

    The kernels build gradients by adding into zero-filled buffers.  A table that is a LEAF whose dense `.grad` already exists
    (zero-filled by `zero_grad(set_to_none=False)` / FusedOptimizer.clip_and_step(zero_grads=True), or holding what earlier backward
    calls of the step left) takes those adds straight into `.grad`, and the Function reports None for it: no zero-filled temporary,
    no engine-side sum of the positive and the negative call's contributions, no AccumulateGrad add -- three whole-table passes and
    launches per table and call (the B = 512 step through autograd is bound by its ~50 launches, not by arithmetic).  The same
    trade as the gradient-accumulation fusion of large-model trainers (the weight gradient written into main_grad, None returned).
    Everything else -- non-leaves (the zero-tail staged tables), leaves without a `.grad` yet, hooks on the tensor, create_graph
    (grad mode on inside backward), set_direct_grad(False) -- gets zero-filled buffers out of ONE allocation and ONE fill launch and
    the usual autograd hand-over.  `None` entries (absent tables) pass through."""
    direct = _DIRECT_GRAD[0] and not torch.is_grad_enabled()
    bufs, rets, fresh, taken = [], [], [], set()
    for t in tables:
        if t is None:
            bufs.append(None); rets.append(None)
        elif direct and _takes_grad_directly(t) and t.grad.data_ptr() not in taken:
            # (one table passed twice keeps ONE direct buffer: the sorted-segment reductions of a large batch add without atomics and
            #  must not meet in the same rows; the second role gets its own buffer and autograd adds it)
            taken.add(t.grad.data_ptr())
            bufs.append(t.grad); rets.append(None)
        else:
            bufs.append(t); rets.append(t); fresh.append(len(bufs) - 1)
    if fresh:
        for k, z in zip(fresh, _zeros_like_many(*[bufs[k] for k in fresh])):
            bufs[k] = rets[k] = z
    return bufs, rets


def _vec(t, n):
    t = t.contiguous()
    if t.dtype != torch.float32 or t.numel() != n:
        raise L.KtupError('expected %d fp32 values, got %s %s' % (n, t.dtype, tuple(t.shape)))
    return t


# ------------------------------------------------------------------------------------------ K1 BPRMF
class _ScoreBprmf(Function):
    @staticmethod
    def forward(ctx, U, I, u, i):
        dev = _dev(_table('user table', U)); _table('item table', I)
        n = u.numel(); u = _ids('u_ids', u, dev); i = _ids('i_ids', i, dev, n)
        score = torch.empty(n, dtype=torch.float32, device=dev)
        L.call('ktup_score_bprmf_fwd', _p(U), U.stride(0), _p(I), I.stride(0), U.shape[1], _p(u), _p(i), n, _p(score), _stream(dev))
        ctx.save_for_backward(U, I, u, i)
        return score

    @staticmethod
    def backward(ctx, gs):
        U, I, u, i = ctx.saved_tensors
        gs = _vec(gs, u.numel())
        (gU, gI), rets = _grad_targets(U, I)
        bws = _seg_ws(L.load().ktup_score_bprmf_bwd_workspace_bytes(u.numel(), U.shape[1], U.shape[0], I.shape[0]), U.device)
        L.call('ktup_score_bprmf_bwd_ws', _p(U), U.stride(0), _p(I), I.stride(0), U.shape[1], _p(u), _p(i), u.numel(), _p(gs),
               _p(gU), _p(gI), U.shape[0], I.shape[0], _p(bws), _stream(U.device))
        return rets[0], rets[1], None, None


def score_bprmf(U, I, u, i):
    """bprmf.py:46-49."""
    return _ScoreBprmf.apply(U, I, u, i)


# ------------------------------------------------------------------------------------------ K2-K4 TransE/H/R
class _ScoreTransE(Function):
    @staticmethod
    def forward(ctx, E, R, h, t, r, l1):
        dev = _dev(_table('entity table', E)); _table('relation table', R)
        n = h.numel(); h = _ids('h', h, dev); t = _ids('t', t, dev, n); r = _ids('r', r, dev, n)
        score = torch.empty(n, dtype=torch.float32, device=dev)
        L.call('ktup_score_transe_fwd', _p(E), E.stride(0), _p(R), R.stride(0), R.shape[0], E.shape[1], _p(h), _p(t), _p(r), n, int(l1),
               _p(score), _stream(dev))
        ctx.save_for_backward(E, R, h, t, r); ctx.l1 = int(l1)
        return score

    @staticmethod
    def backward(ctx, gs):
        E, R, h, t, r = ctx.saved_tensors
        gs = _vec(gs, h.numel())
        (gE, gR), rets = _grad_targets(E, R)
        bws = _seg_ws(L.load().ktup_score_kg_bwd_workspace_bytes(h.numel(), E.shape[1], E.shape[0]), E.device)
        L.call('ktup_score_transe_bwd_ws', _p(E), E.stride(0), _p(R), R.stride(0), E.shape[1], _p(h), _p(t), _p(r), h.numel(),
               ctx.l1, _p(gs), _p(gE), _p(gR), E.shape[0], R.shape[0], _p(bws), _stream(E.device))
        return rets[0], rets[1], None, None, None, None


class _ScoreTransH(Function):
    @staticmethod
    def forward(ctx, E, R, N, h, t, r, l1):
        dev = _dev(_table('entity table', E)); _table('relation table', R); _table('norm table', N)
        n = h.numel(); h = _ids('h', h, dev); t = _ids('t', t, dev, n); r = _ids('r', r, dev, n)
        score = torch.empty(n, dtype=torch.float32, device=dev)
        L.call('ktup_score_transh_fwd', _p(E), E.stride(0), _p(R), R.stride(0), _p(N), N.stride(0), min(R.shape[0], N.shape[0]),
               E.shape[1], _p(h), _p(t), _p(r), n, int(l1), _p(score), _stream(dev))
        ctx.save_for_backward(E, R, N, h, t, r); ctx.l1 = int(l1)
        return score

    @staticmethod
    def backward(ctx, gs):
        E, R, N, h, t, r = ctx.saved_tensors
        gs = _vec(gs, h.numel())
        (gE, gR, gN), rets = _grad_targets(E, R, N)
        bws = _seg_ws(L.load().ktup_score_kg_bwd_workspace_bytes(h.numel(), E.shape[1], E.shape[0]), E.device)
        L.call('ktup_score_transh_bwd_ws', _p(E), E.stride(0), _p(R), R.stride(0), _p(N), N.stride(0), E.shape[1], _p(h), _p(t),
               _p(r), h.numel(), ctx.l1, _p(gs), _p(gE), _p(gR), _p(gN), E.shape[0], min(R.shape[0], N.shape[0]), _p(bws),
               _stream(E.device))
        return rets[0], rets[1], rets[2], None, None, None, None


class _ScoreTransR(Function):
    @staticmethod
    def forward(ctx, E, R, M, h, t, r, l1):
        dev = _dev(_table('entity table', E)); _table('relation table', R); _table('projection table', M)
        n = h.numel(); h = _ids('h', h, dev); t = _ids('t', t, dev, n); r = _ids('r', r, dev, n)
        score = torch.empty(n, dtype=torch.float32, device=dev)
        n_rel = min(R.shape[0], M.shape[0])
        nbytes = L.load().ktup_score_transr_workspace_bytes(n, n_rel)
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=dev) if nbytes else None
        L.call('ktup_score_transr_fwd', _p(E), E.stride(0), _p(R), R.stride(0), _p(M), M.stride(0), n_rel, E.shape[1], _p(h), _p(t),
               _p(r), n, int(l1), _p(score), _p(ws), _stream(dev))
        ctx.save_for_backward(E, R, M, h, t, r); ctx.l1 = int(l1)
        return score

    @staticmethod
    def backward(ctx, gs):
        E, R, M, h, t, r = ctx.saved_tensors
        gs = _vec(gs, h.numel())
        gE, gR, gM = _zeros_like_many(E, R, M)
        n, n_rel = h.numel(), min(R.shape[0], M.shape[0])
        ws = _seg_ws(L.load().ktup_score_transr_bwd_workspace_bytes(n, E.shape[1], E.shape[0], n_rel), E.device)
        L.call('ktup_score_transr_bwd_ws', _p(E), E.stride(0), _p(R), R.stride(0), _p(M), M.stride(0), E.shape[1], _p(h), _p(t),
               _p(r), n, ctx.l1, _p(gs), _p(gE), _p(gR), _p(gM), E.shape[0], n_rel, _p(ws), _stream(E.device))
        return gE, gR, gM, None, None, None, None


def score_transe(E, R, h, t, r, l1):
    """transE.py:51-63."""
    return _ScoreTransE.apply(E, R, h, t, r, l1)


def score_transh(E, R, N, h, t, r, l1):
    """transH.py:58-71 (also KTUP's KG branch, jTransUP.py:144-157)."""
    return _ScoreTransH.apply(E, R, N, h, t, r, l1)


def score_transr(E, R, M, h, t, r, l1):
    """transR.py:65-78."""
    return _ScoreTransR.apply(E, R, M, h, t, r, l1)


# ------------------------------------------------------------------------------------------ K5-K7 TUP / KTUP
# Any -embedding_size (models/base.py:52 takes any integer): the TUP / KTUP kernels read rows as 16-byte chunks, so a width that is
# not a multiple of 4 is staged with a zero tail -- zero coordinates add nothing to a logit, a projection, a distance or a gradient of
# the real ones.  The staging is a torch pad (its backward slices the gradients back): whole tables for the scoring calls, the
# queried user rows and the catalogue for evaluation.  The arithmetic stays in the HIP kernels.
def _pad4(t):
    if t is None or t.shape[1] % 4 == 0:
        return t
    return torch.nn.functional.pad(t, (0, (-t.shape[1]) % 4))


def _pad_users(U, u):
    """(zero-padded rows U[u], ids 0 .. len(u) - 1) for a user table whose width is not a multiple of 4; (U, u) otherwise."""
    if U.shape[1] % 4 == 0:
        return U, u
    u = _ids('u_ids', u, _dev(_table('user table', U)))
    return _pad4(U.index_select(0, u)), torch.arange(u.numel(), dtype=torch.int64, device=U.device)


def pref_workspace(pref, pref_norm, rel=None, norm=None):
    """Mixed, pre-scaled preference tables (ktup_pref_prepare) for the CURRENT table contents.

    Deliberately not cached across calls: tensor identity / version counters do not see `.data` writes
    (trainer.loadEmbedding does exactly that) and allocator address reuse makes pointer keys unsafe.  The prepare
    kernel touches ~24 KB; forward saves the workspace so backward never re-runs it."""
    dev = _dev(_table('pref table', pref)); _table('pref_norm table', pref_norm)
    tabs = [pref, pref_norm] + ([rel, norm] if rel is not None else [])
    for t in tabs:
        _table('preference-side table', t)
        if t.shape != pref.shape or t.stride(0) != pref.stride(0):
            raise L.KtupError('preference / relation tables must share shape and pitch')
    P, d = pref.shape
    nbytes = L.load().ktup_pref_workspace_bytes(d, P)
    if nbytes == 0:
        raise L.KtupError('TUP / KTUP beyond 256 columns: at most 128 preferences (got %d)' % P
                          if d > 256 and d % 4 == 0 else 'TUP / KTUP kernels take rows of whole 16-byte chunks (ops stages other widths with a zero tail); got d=%d' % d)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    L.call('ktup_pref_prepare', _p(pref), _p(pref_norm), _p(rel), _p(norm), pref.stride(0), P, d, _p(ws), _stream(dev))
    return ws


_philox_offset = [0]


def next_philox_offset(count):
    off = _philox_offset[0]
    _philox_offset[0] += int(count)
    return off


class _ScorePref(Function):
    """TUP when E is None, KTUP otherwise."""

    @staticmethod
    def forward(ctx, U, I, E, pref, pref_norm, rel, norm, item2ent, u, i, l1, gumbel_mode, uniform, seed, offset, ent_pad, ws):
        dev = _dev(_table('user table', U)); _table('item table', I)
        n = u.numel(); u = _ids('u_ids', u, dev); i = _ids('i_ids', i, dev, n)
        P, d = pref.shape
        if ws is None:
            ws = pref_workspace(pref, pref_norm, rel, norm)
        if gumbel_mode == GUMBEL_INPUT:
            if uniform is None or tuple(uniform.shape) != (n, P) or uniform.dtype != torch.float32 or uniform.device != dev:
                raise L.KtupError('uniform must be an (n, n_pref) fp32 device tensor')
            uniform = uniform.contiguous()
        elif gumbel_mode == GUMBEL_PHILOX_DEV:     # `uniform` is the device-side stream position {seed, offset}
            if uniform is None or uniform.dtype != torch.int64 or uniform.numel() != 2 or uniform.device != dev or not uniform.is_contiguous():
                raise L.KtupError('GUMBEL_PHILOX_DEV takes a contiguous int64[2] device tensor {seed, offset}')
        else:
            uniform = None
        score = torch.empty(n, dtype=torch.float32, device=dev)
        if E is None:
            L.call('ktup_score_tup_fwd', _p(U), U.stride(0), _p(I), I.stride(0), _p(ws), P, d, _p(u), _p(i), n, int(l1),
                   int(gumbel_mode), _p(uniform), int(seed), int(offset), _p(score), _stream(dev))
        else:
            _table('entity table', E)
            if item2ent.dtype != torch.int32 or item2ent.device != dev or item2ent.numel() < I.shape[0]:
                raise L.KtupError('item2ent must be an int32 device table with one entry per item')
            L.call('ktup_score_ktup_fwd', _p(U), U.stride(0), _p(I), I.stride(0), _p(E), E.stride(0), _p(item2ent), _p(ws), P, d,
                   _p(u), _p(i), n, int(l1), int(gumbel_mode), _p(uniform), int(seed), int(offset), _p(score), _stream(dev))
        ctx.save_for_backward(U, I, E, pref, pref_norm, rel, norm, item2ent, u, i, uniform, ws)
        ctx.cfg = (int(l1), int(gumbel_mode), int(seed), int(offset), int(ent_pad))
        return score

    @staticmethod
    def backward(ctx, gs):
        U, I, E, pref, pref_norm, rel, norm, item2ent, u, i, uniform, ws = ctx.saved_tensors
        l1, gumbel_mode, seed, offset, ent_pad = ctx.cfg
        n = u.numel(); P, d = pref.shape; dev = U.device
        gs = _vec(gs, n)
        # the row tables' gradients go where _grad_targets says (straight into `.grad` when it exists); the mixed tables A = pref + rel,
        # C = pref_norm + norm of KTUP have ONE gradient each for two summands: the kernel builds it in a small zero-filled buffer
        # and one multi-tensor add hands it to the summands that take their gradient directly
        if E is None:
            (gU, gI, gA, gC), rets = _grad_targets(U, I, pref, pref_norm)
            gE, small = None, None
        else:
            (gU, gI, gE), rets = _grad_targets(U, I, E)
            small = (pref, pref_norm, rel, norm)
            direct = _DIRECT_GRAD[0] and not torch.is_grad_enabled() and all(_takes_grad_directly(t) for t in small)
            gA, gC = _zeros_like_many(pref, pref_norm)
        # large batches: per-pair row gradients + reduction by sorted segments instead of float atomics (the library decides:
        # 0 bytes = the atomics path)
        nbytes = L.load().ktup_score_pref_bwd_workspace_bytes(n, d, U.shape[0], I.shape[0])
        bws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=dev) if nbytes else None
        if E is None:
            L.call('ktup_score_tup_bwd_ws', _p(U), U.stride(0), _p(I), I.stride(0), _p(ws), P, d, _p(u), _p(i), n, l1, gumbel_mode,
                   _p(uniform), seed, offset, _p(gs), _p(gU), _p(gI), _p(gA), _p(gC), U.shape[0], I.shape[0], _p(bws), _stream(dev))
            return rets[0], rets[1], None, rets[2], rets[3], None, None, None, None, None, None, None, None, None, None, None, None
        L.call('ktup_score_ktup_bwd_ws', _p(U), U.stride(0), _p(I), I.stride(0), _p(E), E.stride(0), _p(item2ent), ent_pad, _p(ws), P,
               d, _p(u), _p(i), n, l1, gumbel_mode, _p(uniform), seed, offset, _p(gs), _p(gU), _p(gI), _p(gE), _p(gA), _p(gC),
               U.shape[0], I.shape[0], _p(bws), _stream(dev))
        # A = pref + rel and C = pref_norm + norm: the mixed-table gradient goes to both summands
        if direct:
            torch._foreach_add_([t.grad for t in small], [gA, gC, gA, gC])
            return rets[0], rets[1], rets[2], None, None, None, None, None, None, None, None, None, None, None, None, None, None
        return rets[0], rets[1], rets[2], gA, gC, gA.clone(), gC.clone(), None, None, None, None, None, None, None, None, None, None


def score_tup(U, I, pref, pref_norm, u, i, l1, gumbel_mode=GUMBEL_OFF, uniform=None, seed=0, offset=0, ws=None):
    """transUP.py:69-82 (forward) with getPreferences / st_gumbel_softmax fused (transUP.py:105-170).
    `ws`: a pref_workspace(...) of the CURRENT table contents, to share one ktup_pref_prepare between several calls
    (pos / neg batches of a step); None prepares one here."""
    if pref.shape[1] % 4:
        U, I, pref, pref_norm, ws = _pad4(U), _pad4(I), _pad4(pref), _pad4(pref_norm), None
    return _ScorePref.apply(U, I, None, pref, pref_norm, None, None, None, u, i, l1, gumbel_mode, uniform, seed, offset, -1, ws)


def score_ktup(U, I, E, pref, pref_norm, rel, norm, item2ent, u, i, l1, gumbel_mode=GUMBEL_OFF, uniform=None, seed=0,
               offset=0, ent_pad=-1, ws=None):
    """jTransUP.py:122-143 (is_rec branch) with paddingItems replaced by the int32 `item2ent` device table.  `ws` as in
    score_tup."""
    if pref.shape[1] % 4:
        U, I, E, pref, pref_norm, rel, norm, ws = _pad4(U), _pad4(I), _pad4(E), _pad4(pref), _pad4(pref_norm), _pad4(rel), _pad4(norm), None
    return _ScorePref.apply(U, I, E, pref, pref_norm, rel, norm, item2ent, u, i, l1, gumbel_mode, uniform, seed, offset, ent_pad,
                            ws)


# ------------------------------------------------------------------------------------------ K8-K10 losses
class _PairLoss(Function):
    @staticmethod
    def forward(ctx, pos, neg, param, margin):
        dev = _dev(pos)
        n = pos.numel()
        pos = _vec(pos, n); neg = _vec(neg, n)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        L.call('ktup_loss_margin_fwd' if margin else 'ktup_loss_bpr_fwd', _p(pos), _p(neg), n, float(param), _p(loss), _stream(dev))
        ctx.save_for_backward(pos, neg); ctx.cfg = (float(param), bool(margin))
        return loss

    @staticmethod
    def backward(ctx, gloss):
        pos, neg = ctx.saved_tensors
        param, margin = ctx.cfg
        gloss = gloss.contiguous().float()
        gpos, gneg = torch.empty_like(pos), torch.empty_like(neg)
        L.call('ktup_loss_margin_bwd' if margin else 'ktup_loss_bpr_bwd', _p(pos), _p(neg), pos.numel(), param, _p(gloss),
               _p(gpos), _p(gneg), _stream(pos.device))
        return gpos, gneg, None, None


def bpr_loss(pos, neg, target=1.0):
    """utils/loss.py:29-31 -- a MEAN over the batch."""
    return _PairLoss.apply(pos, neg, target, False)


def margin_loss(pos, neg, margin):
    """utils/loss.py:8-16 -- a SUM over the batch."""
    return _PairLoss.apply(pos, neg, margin, True)


class _NormLoss(Function):
    @staticmethod
    def forward(ctx, T, ids):
        dev = _dev(_table('embedding table', T))
        n = T.shape[0] if ids is None else ids.numel()
        ids = None if ids is None else _ids('ids', ids, dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        L.call('ktup_reg_norm_fwd', _p(T), T.stride(0), T.shape[1], _p(ids), n, _p(loss), _stream(dev))
        ctx.save_for_backward(T, ids); ctx.n = n
        return loss

    @staticmethod
    def backward(ctx, gloss):
        T, ids = ctx.saved_tensors
        gT = torch.zeros_like(T)
        gloss = gloss.contiguous().float()
        L.call('ktup_reg_norm_bwd', _p(T), T.stride(0), T.shape[1], _p(ids), ctx.n, _p(gloss), _p(gT), _stream(T.device))
        return gT, None


class _OrthLoss(Function):
    @staticmethod
    def forward(ctx, R, N, ids):
        dev = _dev(_table('rel table', R)); _table('norm table', N)
        if R.shape != N.shape:
            raise L.KtupError('orthogonalLoss: rel and norm tables must have the same shape')
        n = R.shape[0] if ids is None else ids.numel()
        ids = None if ids is None else _ids('ids', ids, dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        L.call('ktup_reg_orth_fwd', _p(R), R.stride(0), _p(N), N.stride(0), R.shape[1], _p(ids), n, _p(loss), _stream(dev))
        ctx.save_for_backward(R, N, ids); ctx.n = n
        return loss

    @staticmethod
    def backward(ctx, gloss):
        R, N, ids = ctx.saved_tensors
        gR, gN = _zeros_like_many(R, N)
        gloss = gloss.contiguous().float()
        L.call('ktup_reg_orth_bwd', _p(R), R.stride(0), _p(N), N.stride(0), R.shape[1], _p(ids), ctx.n, _p(gloss), _p(gR), _p(gN),
               _stream(R.device))
        return gR, gN, None


def norm_loss(table, ids=None):
    """utils/loss.py:21-23 over rows table[ids] (ids None = every row); the gather is fused."""
    return _NormLoss.apply(table, ids)


def orthogonal_loss(rel, norm, ids=None):
    """utils/loss.py:18-19 over row pairs (rel[ids], norm[ids])."""
    return _OrthLoss.apply(rel, norm, ids)


# ------------------------------------------------------------------------------------------ K11-K16 evaluation scores
def _scratch(nbytes, dev):
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)


@torch.no_grad()
def eval_bprmf(U, I, u):
    """bprmf.py:51-54 -> (len(u), n_items)."""
    dev = _dev(_table('user table', U)); _table('item table', I)
    u = _ids('u_ids', u, dev)
    out = torch.empty(u.numel(), I.shape[0], dtype=torch.float32, device=dev)
    L.call('ktup_eval_bprmf_scores', _p(U), U.stride(0), _p(I), I.stride(0), U.shape[1], _p(u), u.numel(), I.shape[0], _p(out),
           out.stride(0), _stream(dev))
    return out


@torch.no_grad()
def eval_transe(E, R, q, r, l1, head, candidates=None):
    """transE.py:65-105 -> (len(q), n_candidates)."""
    dev = _dev(_table('entity table', E)); _table('relation table', R)
    C = E if candidates is None else _table('candidate table', candidates)
    nq = q.numel(); q = _ids('q', q, dev); r = _ids('r', r, dev, nq)
    out = torch.empty(nq, C.shape[0], dtype=torch.float32, device=dev)
    ws = _scratch(L.load().ktup_eval_kg_workspace_bytes(E.shape[1], nq), dev)
    L.call('ktup_eval_transe_scores', _p(E), E.stride(0), _p(R), R.stride(0), E.shape[1], _p(C), C.stride(0), C.shape[0], _p(q),
           _p(r), nq, int(l1), int(head), _p(out), out.stride(0), _p(ws), _stream(dev))
    return out


@torch.no_grad()
def eval_transh(E, R, N, q, r, l1, head, candidates=None):
    """transH.py:73-121 / jTransUP.py:193-247 -> (len(q), n_candidates)."""
    dev = _dev(_table('entity table', E)); _table('relation table', R); _table('norm table', N)
    C = E if candidates is None else _table('candidate table', candidates)
    nq = q.numel(); q = _ids('q', q, dev); r = _ids('r', r, dev, nq)
    out = torch.empty(nq, C.shape[0], dtype=torch.float32, device=dev)
    ws = _scratch(L.load().ktup_eval_kg_workspace_bytes(E.shape[1], nq), dev)
    L.call('ktup_eval_transh_scores', _p(E), E.stride(0), _p(R), R.stride(0), _p(N), N.stride(0), E.shape[1], _p(C), C.stride(0),
           C.shape[0], _p(q), _p(r), nq, int(l1), int(head), _p(out), out.stride(0), _p(ws), _stream(dev))
    return out


KG_TRANSE, KG_TRANSH = 0, 1


_MAX_GOLDS = {}


def _max_golds(gold_off):
    """Largest gold set of a pass's CSR index: one device read per index (the offsets of a pass live as long as the run)."""
    key = (gold_off.data_ptr(), gold_off.numel())
    hit = _MAX_GOLDS.get(key)
    if hit is None or hit[0] is not gold_off:
        if len(_MAX_GOLDS) > 64:
            _MAX_GOLDS.clear()
        hit = _MAX_GOLDS[key] = (gold_off, int((gold_off[1:] - gold_off[:-1]).max().item()) if gold_off.numel() > 1 else 0)
    return hit[1]


@torch.no_grad()
def eval_kg_ranks(E, R, N, q, r, l1, head, descending, gold_off, gold_ids, filt_off=None, filt_ids=None, candidates=None, chunk=512,
                  fused=None):
    """A whole link-prediction pass in one call (ktup_eval_kg_ranks): the filtered 0-based rank of every gold entry of every key
    (q[i], r[i]) among all candidates -- K12 (N is None) or K13 scores `chunk` keys at a time + K18, the loop over the batches
    under the C ABI.  gold_off / filt_off: the pass's CSR offsets (len(q) + 1, absolute).  -> int32 [gold_off[-1]]; -1 = a gold
    id that is itself filtered (the reference's walk never reaches it)."""
    dev = _dev(_table('entity table', E)); _table('relation table', R)
    if N is not None:
        _table('norm table', N)
    C = E if candidates is None else _table('candidate table', candidates)
    nq = q.numel(); q = _ids('q', q, dev); r = _ids('r', r, dev, nq)
    if filt_ids is not None and filt_ids.numel() == 0:       # an empty id list has no storage to point at: same as no filter
        filt_off = filt_ids = None
    if gold_off.numel() != nq + 1 or (filt_off is not None and filt_off.numel() != nq + 1):
        raise L.KtupError('eval_kg_ranks: CSR offsets need len(q) + 1 entries')
    n_gold = gold_ids.numel()
    ranks = torch.empty(max(n_gold, 1), dtype=torch.int32, device=dev)
    if fused is not False and nq > 0 and n_gold > 0:
        # the pass without the score matrix (ktup_eval_kg_ranks_fused): ranks from counts formed where the scores are made -- the
        # matrix-core sweep for squared L2 at its widths, the pair kernels' COUNT form for L1 / other widths / large gold sets
        mg = _max_golds(gold_off)
        lib = L.load()
        model, n_rel = (KG_TRANSE, 0) if N is None else (KG_TRANSH, min(R.shape[0], N.shape[0]))
        if lib.ktup_eval_kg_ranks_fused_supported(model, E.shape[1], int(bool(l1)), mg):
            n_filt = 0 if filt_ids is None else filt_ids.numel()
            fws = _scratch(lib.ktup_eval_kg_ranks_fused_workspace_bytes(model, E.shape[1], nq, n_gold, n_filt, C.shape[0], n_rel), dev)
            try:
                L.call('ktup_eval_kg_ranks_fused', model, _p(E), E.stride(0), _p(R), R.stride(0), _p(N),
                       0 if N is None else N.stride(0), n_rel, E.shape[1], _p(C), C.stride(0), C.shape[0], _p(q), _p(r), nq, int(bool(l1)), int(head),
                       int(bool(descending)), _p(filt_off), _p(filt_ids), n_filt, _p(gold_off), _p(gold_ids), n_gold, mg, _p(ranks), _p(fws),
                       _stream(dev))
                return ranks
            except L.KtupError as e:
                if e.code != L.ERR_UNSUPPORTED:        # a shape the pass declines (rows too wide for its LDS tile): the chunked route below
                    raise
    chunk = max(1, min(int(chunk), max(nq, 1)))
    ws = _scratch(L.load().ktup_eval_kg_ranks_workspace_bytes(E.shape[1], C.shape[0], chunk), dev)
    L.call('ktup_eval_kg_ranks', KG_TRANSE if N is None else KG_TRANSH, _p(E), E.stride(0), _p(R), R.stride(0), _p(N),
           0 if N is None else N.stride(0), E.shape[1], _p(C), C.stride(0), C.shape[0], _p(q), _p(r), nq, int(l1), int(head),
           int(bool(descending)), _p(filt_off), _p(filt_ids), _p(gold_off), _p(gold_ids), _p(ranks), chunk, _p(ws), _stream(dev))
    return ranks


@torch.no_grad()
def eval_kg_ranks_transr(E, R, M, q, r, l1, head, descending, gold_off, gold_ids, filt_off=None, filt_ids=None, ents=None, chunk=512):
    """eval_kg_ranks for TransR (ktup_eval_kg_ranks_transr): K14 per chunk of keys against the entity side `ents`
    (eval_transr_entities, once per pass) + K18, the loop under the C ABI."""
    dev = _dev(_table('entity table', E)); _table('relation table', R); _table('projection table', M)
    nq = q.numel(); q = _ids('q', q, dev); r = _ids('r', r, dev, nq)
    if filt_ids is not None and filt_ids.numel() == 0:
        filt_off = filt_ids = None
    if gold_off.numel() != nq + 1 or (filt_off is not None and filt_off.numel() != nq + 1):
        raise L.KtupError('eval_kg_ranks_transr: CSR offsets need len(q) + 1 entries')
    if ents is not None and (ents.shape != (E.shape[0], E.shape[1], R.shape[0]) or ents.l1 != bool(l1)):
        raise L.KtupError('prepared entity side does not match the tables / distance kind')
    ranks = torch.empty(max(gold_ids.numel(), 1), dtype=torch.int32, device=dev)
    chunk = max(1, min(int(chunk), max(nq, 1)))
    ws = _scratch(L.load().ktup_eval_kg_ranks_transr_workspace_bytes(E.shape[1], E.shape[0], R.shape[0], chunk), dev)
    L.call('ktup_eval_kg_ranks_transr', _p(E), E.stride(0), _p(R), R.stride(0), _p(M), M.stride(0), E.shape[1], E.shape[0], R.shape[0],
           _p(None if ents is None else ents.ws), _p(q), _p(r), nq, int(l1), int(head), int(bool(descending)), _p(filt_off), _p(filt_ids),
           _p(gold_off), _p(gold_ids), _p(ranks), chunk, _p(ws), _stream(dev))
    return ranks


class PreparedEntities(object):
    """Entity side of K14 for one evaluation pass (ktup_eval_transr_prepare): valid while the tables it was built from do not
    change, for the distance kind it was built for."""

    def __init__(self, ws, shape, l1):
        self.ws, self.shape, self.l1 = ws, shape, bool(l1)


@torch.no_grad()
def eval_transr_entities(E, M, n_rel, l1):
    """|M_r e|^2 (squared L2 on the matrix cores) or M_r e (L1, other widths) for every relation and entity, once per pass."""
    dev = _dev(_table('entity table', E)); _table('projection table', M)
    ws = _scratch(L.load().ktup_eval_transr_entities_workspace_bytes(E.shape[1], E.shape[0], n_rel), dev)
    L.call('ktup_eval_transr_prepare', _p(E), E.stride(0), _p(M), M.stride(0), E.shape[1], E.shape[0], n_rel, int(l1), _p(ws), _stream(dev))
    return PreparedEntities(ws, (E.shape[0], E.shape[1], n_rel), l1)


@torch.no_grad()
def eval_transr(E, R, M, q, r, l1, head, ents=None):
    """transR.py:80-128 -> (len(q), n_entities).  `ents`: eval_transr_entities(...) of the same tables (one per evaluation pass)."""
    dev = _dev(_table('entity table', E)); _table('relation table', R); _table('projection table', M)
    nq = q.numel(); q = _ids('q', q, dev); r = _ids('r', r, dev, nq)
    if ents is not None and (ents.shape != (E.shape[0], E.shape[1], R.shape[0]) or ents.l1 != bool(l1)):
        raise L.KtupError('prepared entity side does not match the tables / distance kind')
    out = torch.empty(nq, E.shape[0], dtype=torch.float32, device=dev)
    ws = _scratch(L.load().ktup_eval_transr_workspace_bytes(E.shape[1], nq, E.shape[0], R.shape[0]), dev)
    L.call('ktup_eval_transr_scores', _p(E), E.stride(0), _p(R), R.stride(0), _p(M), M.stride(0), E.shape[1], E.shape[0], R.shape[0],
           _p(q), _p(r), nq, int(l1), int(head), _p(out), out.stride(0), _p(ws), _p(None if ents is None else ents.ws), _stream(dev))
    return out


class PreparedItems(object):
    """Item side of K15 / K16 for one evaluation pass: the prepared preference tables, the item tables they go with and -- built on
    first use, by the per-batch score route only -- the per-item projections (ktup_eval_pref_items_prepare).  Valid as long as
    the tables it was built from do not change."""

    def __init__(self, pws, I, E, item2ent, P, d):
        self.pws, self.I, self.E, self.item2ent, self.n_items, self.P, self.d = pws, I, E, item2ent, I.shape[0], P, d
        self._items_ws = None

    @property
    def items_ws(self):
        if self._items_ws is None:
            dev, I, E = self.I.device, self.I, self.E
            self._items_ws = _scratch(L.load().ktup_eval_pref_items_workspace_bytes(self.d, self.P, self.n_items), dev)
            L.call('ktup_eval_pref_items_prepare', _p(I), I.stride(0), _p(E), 0 if E is None else E.stride(0), _p(self.item2ent),
                   _p(self.pws), self.P, self.d, self.n_items, _p(self._items_ws), _stream(dev))
        return self._items_ws


@torch.no_grad()
def eval_pref_items(I, E, pref, pref_norm, rel, norm, item2ent):
    I, E, pref, pref_norm, rel, norm = (_pad4(t) for t in (I, E, pref, pref_norm, rel, norm))
    dev = _dev(_table('item table', I))
    ni = I.shape[0]
    P, d = pref.shape
    pws = pref_workspace(pref, pref_norm, rel, norm)
    if E is not None:
        _table('entity table', E)
        if item2ent.dtype != torch.int32 or item2ent.device != dev or item2ent.numel() != ni:
            raise L.KtupError('item2ent must be an int32 device table with one entry per item row')
    return PreparedItems(pws, I, E, item2ent if E is not None else None, P, d)


@torch.no_grad()
def _eval_pref(U, I, E, pref, pref_norm, rel, norm, item2ent, u, l1, gumbel_mode, uniform, seed, offset, items=None):
    if pref.shape[1] % 4:
        U, u = _pad_users(U, u)
        I, E, pref, pref_norm, rel, norm = (_pad4(t) for t in (I, E, pref, pref_norm, rel, norm))
    dev = _dev(_table('user table', U)); _table('item table', I)
    u = _ids('u_ids', u, dev)
    nq, ni = u.numel(), I.shape[0]
    P, d = pref.shape
    if items is not None and d > 256:            # the one-wave-per-pair route (ktup_score_pref_row.hip) stages no item side
        if (items.n_items, items.P, items.d) != (ni, P, d):
            raise L.KtupError('prepared item side does not match the tables')
        items = None
    if items is not None:                        # the pass prepared the item side once: users only here
        if (items.n_items, items.P, items.d) != (ni, P, d):
            raise L.KtupError('prepared item side does not match the tables')
        if gumbel_mode == GUMBEL_INPUT:
            if uniform is None or tuple(uniform.shape) != (nq, ni, P) or uniform.dtype != torch.float32 or uniform.device != dev:
                raise L.KtupError('uniform must be an (n_users, n_items, n_pref) fp32 device tensor')
            uniform = uniform.contiguous()
        else:
            uniform = None
        out = torch.empty(nq, ni, dtype=torch.float32, device=dev)
        ws = _scratch(L.load().ktup_eval_pref_workspace_bytes(d, P, nq, 0), dev)
        L.call('ktup_eval_pref_scores_prepared', _p(U), U.stride(0), _p(items.pws), P, d, _p(u), nq, ni, int(l1), int(gumbel_mode),
               _p(uniform), int(seed), int(offset), _p(out), out.stride(0), _p(items.items_ws), _p(ws), _stream(dev))
        return out
    pws = pref_workspace(pref, pref_norm, rel, norm)
    if gumbel_mode == GUMBEL_INPUT:
        if uniform is None or tuple(uniform.shape) != (nq, ni, P) or uniform.dtype != torch.float32 or uniform.device != dev:
            raise L.KtupError('uniform must be an (n_users, n_items, n_pref) fp32 device tensor')
        uniform = uniform.contiguous()
    else:
        uniform = None
    if E is not None:
        _table('entity table', E)
        if item2ent.dtype != torch.int32 or item2ent.device != dev or item2ent.numel() != ni:
            raise L.KtupError('item2ent must be an int32 device table with one entry per item row')
    out = torch.empty(nq, ni, dtype=torch.float32, device=dev)
    ws = _scratch(L.load().ktup_eval_pref_workspace_bytes(d, P, nq, ni), dev)
    L.call('ktup_eval_pref_scores', _p(U), U.stride(0), _p(I), I.stride(0), _p(E), 0 if E is None else E.stride(0), _p(item2ent),
           _p(pws), P, d, _p(u), nq, ni, int(l1), int(gumbel_mode), _p(uniform), int(seed), int(offset), _p(out), out.stride(0),
           _p(ws), _stream(dev))
    return out


@torch.no_grad()
def eval_pref_topk(U, u, items, l1, topn, filt_off=None, filt_ids=None, with_scores=False):
    """Scores + filtered top-n of a whole evaluation pass in one sweep (ktup_eval_pref_topk): every user of `u` against the
    item tables of `items`, no (users x items) matrix.  Squared L2 at d in {64, 100, 128}: the preference-space pass on the matrix cores;
    L1 and other widths: the pair kernel's arithmetic swept with the top-n in its epilogue (the per-batch scores' bits).  -> int32
    (len(u), topn) ids (-1 padded) [, scores], or None when no sweep covers the shape (topn > 16, very wide rows): keep eval_tup /
    eval_ktup + topk_filtered."""
    U, u = _pad_users(U, u)
    dev = _dev(_table('user table', U))
    u = _ids('u_ids', u, dev)
    nq, d, P = u.numel(), items.d, items.P
    if not (0 < topn <= 16) or nq == 0 or d > 256:
        return None
    if l1 or d not in (64, 100, 128) or not L.get_option('eval_mc'):
        # no preference-space pass for this shape: the pair kernel's arithmetic with the top-n in its epilogue (or None: per-batch calls)
        return eval_pref_topk_hard(U, u, items, l1, topn, GUMBEL_OFF, None, 0, 0, filt_off, filt_ids, with_scores)
    if filt_ids is not None and filt_ids.numel() == 0:
        filt_off = filt_ids = None
    top = torch.empty(nq, topn, dtype=torch.int32, device=dev)
    ts = torch.empty(nq, topn, dtype=torch.float32, device=dev) if with_scores else None
    ws = _scratch(L.load().ktup_eval_pref_topk_workspace_bytes(d, P, nq, items.n_items, topn), dev)
    I, E = items.I, items.E
    L.call('ktup_eval_pref_topk', _p(U), U.stride(0), _p(I), I.stride(0), _p(E), 0 if E is None else E.stride(0), _p(items.item2ent),
           _p(items.pws), P, d, _p(u), nq, items.n_items, 0, _p(filt_off), _p(filt_ids), int(topn), _p(top), _p(ts), _p(ws), _stream(dev))
    return (top, ts) if with_scores else top


@torch.no_grad()
def eval_pref_topk_hard(U, u, items, l1, topn, gumbel_mode, uniform=None, seed=0, offset=0, filt_off=None, filt_ids=None, with_scores=False):
    """The hard (ST-Gumbel) gate's whole evaluation pass in one sweep (ktup_eval_pref_topk_hard): the scores of eval_tup / eval_ktup
    for the same noise source over ALL users of `u` at once (pair (b, j) draws at ((b n_items + j) P + p) + offset), L1 or squared
    L2, with the filtered top-n taken where the scores are made.  -> int32 (len(u), topn) ids (-1 padded) [, scores]; None when
    topn > 16 or the library declines the shape (keep eval_* + topk_filtered)."""
    U, u = _pad_users(U, u)
    dev = _dev(_table('user table', U))
    u = _ids('u_ids', u, dev)
    nq, d, P, ni = u.numel(), items.d, items.P, items.n_items
    if not (0 < topn <= 16) or nq == 0 or P > 32 or d > 256:
        return None
    if gumbel_mode == GUMBEL_INPUT:
        if uniform is None or tuple(uniform.shape) != (nq, ni, P) or uniform.dtype != torch.float32 or uniform.device != dev:
            raise L.KtupError('uniform must be an (n_users, n_items, n_pref) fp32 device tensor')
        uniform = uniform.contiguous()
    elif gumbel_mode not in (GUMBEL_PHILOX, GUMBEL_OFF):
        raise L.KtupError('eval_pref_topk_hard takes GUMBEL_INPUT, GUMBEL_PHILOX or GUMBEL_OFF (the soft gate scored pair by pair)')
    else:
        uniform = None
    if filt_ids is not None and filt_ids.numel() == 0:
        filt_off = filt_ids = None
    top = torch.empty(nq, topn, dtype=torch.int32, device=dev)
    ts = torch.empty(nq, topn, dtype=torch.float32, device=dev) if with_scores else None
    ws = _scratch(L.load().ktup_eval_pref_topk_hard_workspace_bytes(d, P, nq, ni, topn), dev)
    I, E = items.I, items.E
    try:
        L.call('ktup_eval_pref_topk_hard', _p(U), U.stride(0), _p(I), I.stride(0), _p(E), 0 if E is None else E.stride(0), _p(items.item2ent),
               _p(items.pws), P, d, _p(u), nq, ni, int(l1), int(gumbel_mode), _p(uniform), int(seed), int(offset), _p(filt_off), _p(filt_ids),
               int(topn), _p(top), _p(ts), _p(ws), _stream(dev))
    except L.KtupError as e:
        if e.code == L.ERR_UNSUPPORTED:          # a shape the sweep does not cover (a catalogue split whose filter bitmap passes the LDS):
            return None                          # the caller keeps the per-batch calls
        raise
    return (top, ts) if with_scores else top


def eval_tup(U, I, pref, pref_norm, u, l1, gumbel_mode=GUMBEL_OFF, uniform=None, seed=0, offset=0, items=None):
    """transUP.py:84-102 -> (len(u), n_items).  `items`: eval_pref_items(...) of the same tables (one per evaluation pass)."""
    return _eval_pref(U, I, None, pref, pref_norm, None, None, None, u, l1, gumbel_mode, uniform, seed, offset, items)


def eval_ktup(U, I, E, pref, pref_norm, rel, norm, item2ent, u, l1, gumbel_mode=GUMBEL_OFF, uniform=None, seed=0, offset=0,
              items=None):
    """jTransUP.py:163-191 -> (len(u), n_items)."""
    return _eval_pref(U, I, E, pref, pref_norm, rel, norm, item2ent, u, l1, gumbel_mode, uniform, seed, offset, items)


# ------------------------------------------------------------------------------------------ K17-K18 ranking
@torch.no_grad()
def topk_filtered(scores, descending, topn, filt_off=None, filt_ids=None, with_scores=False):
    """First `topn` unfiltered candidate ids per row (ascending score, ties -> lower id); int32 (nq, topn), -1 padded."""
    dev = _dev(scores)
    if scores.dtype != torch.float32 or scores.dim() != 2 or scores.stride(1) != 1:
        raise L.KtupError('scores must be a 2-D fp32 device matrix with unit inner stride')
    nq, nc = scores.shape
    if filt_ids is not None and filt_ids.numel() == 0:       # an empty id list has no storage to point at: same as no filter
        filt_off = filt_ids = None
    top = torch.empty(nq, topn, dtype=torch.int32, device=dev)
    ts = torch.empty(nq, topn, dtype=torch.float32, device=dev) if with_scores else None
    L.call('ktup_eval_topk_filtered', _p(scores), scores.stride(0), nq, nc, int(bool(descending)), _p(filt_off), _p(filt_ids),
           int(topn), _p(top), _p(ts), _stream(dev))
    return (top, ts) if with_scores else top


@torch.no_grad()
def gold_ranks(scores, descending, gold_off, gold_ids, filt_off=None, filt_ids=None):
    """0-based filtered rank of every gold entry (CSR); -1 where the gold id is itself filtered."""
    dev = _dev(scores)
    if scores.dtype != torch.float32 or scores.dim() != 2 or scores.stride(1) != 1:
        raise L.KtupError('scores must be a 2-D fp32 device matrix with unit inner stride')
    nq, nc = scores.shape
    if filt_ids is not None and filt_ids.numel() == 0:
        filt_off = filt_ids = None
    ranks = torch.empty(gold_ids.numel(), dtype=torch.int32, device=dev)
    L.call('ktup_eval_gold_ranks', _p(scores), scores.stride(0), nq, nc, int(bool(descending)), _p(filt_off), _p(filt_ids),
           _p(gold_off), _p(gold_ids), _p(ranks), _stream(dev))
    return ranks


@torch.no_grad()
def gold_rank_counts(local_scores, cand_lo, descending, gold_off, gold_ids, gold_scores, filt_off=None, filt_ids=None, cand_stride=1):
    """Per gold entry: this candidate shard's unfiltered non-gold candidates ordered before it (ktup_eval_gold_rank_counts);
    all-reduce(sum) over the shards gives the rank (negative -> the gold is itself filtered -> -1).  Local candidate j has the global
    id cand_lo + cand_stride * j."""
    dev = _dev(local_scores)
    if local_scores.dtype != torch.float32 or local_scores.dim() != 2 or (local_scores.shape[1] and local_scores.stride(1) != 1):
        raise L.KtupError('scores must be a 2-D fp32 device matrix with unit inner stride')
    nq, nl = local_scores.shape
    if filt_ids is not None and filt_ids.numel() == 0:
        filt_off = filt_ids = None
    counts = torch.empty(gold_ids.numel(), dtype=torch.int32, device=dev)
    L.call('ktup_eval_gold_rank_counts_strided', _p(local_scores) if nl else None, local_scores.stride(0) if nl else 0, nq, nl, int(cand_lo),
           int(cand_stride), int(bool(descending)), _p(filt_off), _p(filt_ids), _p(gold_off), _p(gold_ids), _p(gold_scores.contiguous()), _p(counts),
           _stream(dev))
    return counts


@torch.no_grad()
def rec_metrics(top_ids, gold_off, gold_ids):
    """(nq, 5) float64 device tensor of (f1, p, r, hit, ndcg) per ranked list (K18b); gold ids ascending per query."""
    dev = _dev(top_ids)
    if top_ids.dtype != torch.int32 or top_ids.dim() != 2 or not top_ids.is_contiguous():
        raise L.KtupError('top_ids must be a contiguous (nq, topn) int32 device tensor')
    nq, topn = top_ids.shape
    out = torch.empty(nq, 5, dtype=torch.float64, device=dev)
    L.call('ktup_eval_rec_metrics', _p(top_ids), nq, topn, _p(gold_off), _p(gold_ids), _p(out), _stream(dev))
    return out


# ------------------------------------------------------------------------------------------ sharded-table exchange halves
@torch.no_grad()
def dedupe(ids):
    """Distinct ids of a batch without a host sync (ktup_shard_dedupe): -> (uniq, inverse) with uniq padded to len(ids) with -1
    (no particular order among the distinct ids) and inverse[e] = position of ids[e] in uniq."""
    dev = _dev(ids)
    ids = _ids('ids', ids, dev)
    n = ids.numel()
    uniq = torch.empty(n, dtype=torch.int64, device=dev)
    inverse = torch.empty(n, dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    nbytes = L.load().ktup_shard_dedupe_workspace_bytes(n)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev) if nbytes else None
    L.call('ktup_shard_dedupe', _p(ids), n, _p(uniq), _p(inverse), _p(count), _p(ws), _stream(dev))
    return uniq, inverse


@torch.no_grad()
def pack_rows(table, ids):
    """out[k] = table[ids[k]] (owner side of the all-to-all lookup)."""
    dev = _dev(_table('table shard', table))
    ids = _ids('ids', ids, dev)
    out = torch.empty(ids.numel(), table.shape[1], dtype=torch.float32, device=dev)
    L.call('ktup_shard_pack_rows', _p(table), table.stride(0), table.shape[1], _p(ids), ids.numel(), _p(out), out.stride(0), _stream(dev))
    return out


@torch.no_grad()
def unpack_rows_add(rows, ids, gtable):
    """gtable[ids[k]] += rows[k] (owner side of the gradient return)."""
    dev = _dev(_table('row gradients', rows)); _table('shard gradient', gtable)
    ids = _ids('ids', ids, dev, rows.shape[0])
    L.call('ktup_shard_unpack_rows_add', _p(rows), rows.stride(0), rows.shape[1], _p(ids), ids.numel(), _p(gtable), gtable.stride(0),
           _stream(dev))
    return gtable


@torch.no_grad()
def segment_reduce_rows(G, ids, gtable, sign_split=None, map2=None, pad2=-1, gtable2=None):
    """gtable[ids[e]] += (+/-) G[e mod len(G)] by sorted segments (ktup_segment_reduce_rows): the large-batch / hot-row
    alternative to float atomics.  len(ids) is len(G) or 2 len(G); entries from `sign_split` on are subtracted."""
    dev = _dev(_table('row gradients', G)); _table('table gradient', gtable)
    n_src, d = G.shape
    ids = _ids('ids', ids, dev)
    m = ids.numel()
    if m not in (n_src, 2 * n_src):
        raise L.KtupError('segment_reduce_rows: len(ids) must be len(G) or 2 len(G)')
    if map2 is not None:
        _table('second table gradient', gtable2)
        if map2.dtype != torch.int32 or map2.device != dev or map2.numel() < gtable.shape[0]:
            raise L.KtupError('map2 must be an int32 device table with one entry per row of gtable')
    nbytes = L.load().ktup_segment_workspace_bytes(m, gtable.shape[0])
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=dev)
    L.call('ktup_segment_reduce_rows', _p(G), G.stride(0), d, n_src, _p(ids), m, m if sign_split is None else int(sign_split),
           gtable.shape[0], _p(gtable), gtable.stride(0), _p(map2), int(pad2), _p(gtable2), 0 if gtable2 is None else gtable2.stride(0),
           _p(ws), _stream(dev))
    return gtable


@torch.no_grad()
def grad_sumsq(tensors, out=None):
    """Sum of squares of a list of dense fp32 device tensors -> one device double (K20's first launch, ktup_optim_gradnorm);
    no host sync.  Tensors must be contiguous and 16-byte aligned."""
    import ctypes
    tensors = [t for t in tensors if t is not None and t.numel()]
    dev = _dev(tensors[0]) if tensors else (out.device if out is not None else torch.device('cuda', torch.cuda.current_device()))
    if out is None:
        out = torch.zeros(1, dtype=torch.float64, device=dev)
    if not tensors:
        return out.zero_()
    for t in tensors:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
            raise L.KtupError('grad_sumsq needs contiguous fp32 tensors on one device')
    n = len(tensors)
    grads = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
    sizes = (ctypes.c_int64 * n)(*[t.numel() for t in tensors])
    L.call('ktup_optim_gradnorm', n, grads, sizes, _p(out), _stream(dev))
    return out


SPARSE_KINDS = {'sgd': 0, 'adagrad': 1}


@torch.no_grad()
def sparse_step(kind, table, state, ids, grows, lr, eps=1e-10, sumsq=None, max_norm=0.0):
    """Owner-side row-sparse optimizer step (ktup_shard_sparse_step): rows `ids` (unique) of `table` (and of the Adagrad
    `state`) are updated in place with the combined row gradients `grows`; clip coefficient from the device double `sumsq`."""
    dev = _dev(_table('table shard', table)); _table('row gradients', grows)
    ids = _ids('ids', ids, dev, grows.shape[0])
    k = SPARSE_KINDS[kind]
    if k == 1:
        _table('optimizer state', state)
        if state.shape != table.shape:
            raise L.KtupError('Adagrad state must have the shape of the table shard')
    L.call('ktup_shard_sparse_step', k, _p(table), table.stride(0), _p(state) if k == 1 else None, state.stride(0) if k == 1 else 0,
           table.shape[1], _p(ids), ids.numel(), _p(grows), grows.stride(0), float(lr), float(eps), _p(sumsq),
           float(max_norm) if sumsq is not None else 0.0, _stream(dev))
    return table
