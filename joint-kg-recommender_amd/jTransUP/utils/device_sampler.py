"""On-device negative samplers (K19) with the constraints of the reference's host samplers (jTransUP/utils/data.py:12-85).

The filter structures live on the device for the whole run: one bit per (user, item) for "rated in any split", and the
sorted 64-bit keys of every known (h, r, t).  Draws are counter-based (Philox) and batch uniqueness is resolved by
deterministic rounds (csrc/ktup_sample.hip), so a (seed, offset) pair reproduces a batch on every rank, whatever the timing.

The kernels never emit an out-of-range id: when no admissible candidate exists (e.g. more rows than admissible items under
unique_in_batch) the row gets an in-range stand-in and a device-side counter is bumped; `check()` (one small sync; the drivers
call it before every evaluation and at the end of training) raises on a non-zero counter."""
import numpy as np
import torch

from jTransUP.hip import lib as L
from jTransUP.hip.ops import _p, _stream

TRIES = 4096   # counter stride per row inside the kernels
KG_STREAM = 1 << 62


class DeviceSampler(object):
    def __init__(self, device, seed=0):
        self.device = torch.device(device)
        self.seed = int(seed) & (2 ** 63 - 1)
        # one Philox counter per sampler kind (the kg stream starts far away from the rec stream), so that the order in which
        # rec and kg batches are drawn -- which prefetching changes -- does not change the draws
        self.offsets = {'rec': 0, 'kg': KG_STREAM}
        self.bitmap = self.keys = None
        self.n_items = self.n_ent = self.n_rel = 0
        self._ws, self._ws_items = None, -1
        self.fail = torch.zeros(1, dtype=torch.int32, device=self.device)     # rows whose constraints could not be met
        # device copy of the Philox counter for the feed launches (ktup_feed_*: batch + negatives in one graph-replayable
        # launch, utils/fast_train.py fed_step); `offset` stays authoritative on the host, both advance in lockstep
        self.offset_dev = torch.tensor([0, KG_STREAM], dtype=torch.int64, device=self.device)      # [rec, kg]
        self._dev_ok = {'rec': True, 'kg': True}

    def check(self):
        """Raise if any draw since the last check had no admissible candidate (syncs the device once)."""
        bad = int(self.fail.item())
        if bad:
            self.fail.zero_()
            raise L.KtupError('%d negative draws had no admissible candidate (batch larger than the admissible items under '
                              'unique_in_batch, or a user / (h, r) that is connected to everything)' % bad)

    # ---- filter structures
    def set_rating_dicts(self, user_total, item_total, all_dicts):
        """all_dicts: [{user: set(items)}, ...] (train + every eval split), as the drivers build them."""
        words = (item_total + 31) // 32
        bits = np.zeros((user_total, words), dtype=np.uint32)
        for dic in all_dicts or []:
            for u, items in dic.items():
                it = np.fromiter(items, dtype=np.int64)
                np.bitwise_or.at(bits[u], it >> 5, (np.uint32(1) << (it & 31).astype(np.uint32)))
        self.bitmap = torch.from_numpy(bits.view(np.int32)).to(self.device) if all_dicts is not None else None
        self.n_items, self.words = item_total, words

    def set_triples(self, entity_total, relation_total, triple_lists):
        """triple_lists: iterable of [(h, t, r), ...] lists (train + every eval split; tail before relation, like the files)."""
        self.n_ent, self.n_rel = entity_total, relation_total
        if triple_lists is None:
            self.keys = None
            return
        arr = np.concatenate([np.asarray(tl, dtype=np.uint64).reshape(-1, 3) for tl in triple_lists if len(tl)])
        keys = (arr[:, 0] * np.uint64(relation_total) + arr[:, 2]) * np.uint64(entity_total) + arr[:, 1]
        self.keys = torch.from_numpy(np.unique(keys).view(np.int64)).to(self.device)

    # ---- draws
    @property
    def offset(self):
        """Draw counters consumed so far, over both streams."""
        return self.offsets['rec'] + self.offsets['kg'] - KG_STREAM

    def _advance(self, n, kind):
        off = self.offsets[kind]
        self.offsets[kind] += int(n) * TRIES
        self._dev_ok[kind] = False
        return off

    def fed(self, n, kind):
        """Account for the n draws a ktup_feed_* launch of `kind` is about to make from its slot of `offset_dev`."""
        if not self._dev_ok[kind]:
            self.offset_dev[0 if kind == 'rec' else 1] = self.offsets[kind]
            self._dev_ok[kind] = True
        self.offsets[kind] += int(n) * TRIES

    def rec_workspace(self):
        if self._ws is None or self._ws_items != self.n_items:     # batch-uniqueness scratch: reset by the entry points themselves
            nbytes = L.load().ktup_negsample_rec_workspace_bytes(self.n_items)
            # all-ones: what ktup_feed_rec expects and leaves behind (ktup_negsample_rec resets it itself on every call)
            self._ws, self._ws_items = torch.full(((nbytes + 7) // 8,), -1, dtype=torch.int64, device=self.device), self.n_items
        return self._ws

    @torch.no_grad()
    def sample_rec(self, u, pos_i, unique_in_batch=True):
        """-> negative item per (u, pos_i) row, always a valid item id; `check()` reports rows whose constraints had no solution."""
        n = u.numel()
        neg = torch.empty(n, dtype=torch.int64, device=self.device)
        ws = self.rec_workspace()
        L.call('ktup_negsample_rec', _p(u.contiguous()), _p(pos_i.contiguous()), n, self.n_items, _p(self.bitmap),
               self.words if self.bitmap is not None else 0, self.seed, self._advance(n, 'rec'), int(unique_in_batch), _p(neg), _p(ws),
               _p(self.fail), _stream(self.device))
        return neg

    @torch.no_grad()
    def sample_kg(self, h, t, r):
        """-> (neg_h, neg_t): exactly one side corrupted per triple."""
        n = h.numel()
        nh = torch.empty(n, dtype=torch.int64, device=self.device)
        nt = torch.empty(n, dtype=torch.int64, device=self.device)
        L.call('ktup_negsample_kg', _p(h.contiguous()), _p(t.contiguous()), _p(r.contiguous()), n, self.n_ent, self.n_rel, _p(self.keys),
               0 if self.keys is None else self.keys.numel(), self.seed, self._advance(n, 'kg'), _p(nh), _p(nt), _p(self.fail),
               _stream(self.device))
        return nh, nt
