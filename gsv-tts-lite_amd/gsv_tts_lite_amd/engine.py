"""Multi-GPU continuous-batching engine: one process per GPU, utterances dealt on demand, no data-path collective.

What it distributes is the reference's own work queue.  `Text2SemanticDecoder.infer_batched`
(gsv_tts/GPT_SoVITS/GPT/t2s_model.py:555-734) keeps B slots busy and refills a finished slot from the head of
the request list (:696-722); `TTS.infer_batched` (gsv_tts/TTS.py:616-633) builds that list from every cut segment
of every text, balances the vocoder batches (:705-720) and puts the audio back in input order (:820-865).
Utterances are independent through GPT and vocoder, so scale-out is: weights replicated, every rank runs the same
slot loop over ITS slots, and the only shared thing is the head of the queue.

  * dealing   requests are sorted longest-first (expected cost = phonemes, LPT order) and ranks pull chunks of a few
              from ONE shared cursor when a slot frees up -- an atomic counter in the process group's store (a host
              round trip per chunk, off the GPU's critical path; `store.add` is the c10d primitive for exactly this).
              A rank whose utterances turn out short simply comes back sooner: dynamic dealing, no length oracle.
              The static snake partition (`scheduler.shard_indices`) remains as the store-less fallback.
  * speakers  `ge`, prompt tokens, prompt phonemes / BERT features exist only on the rank that ran the
              reference-audio models.  `SpeakerBook.sync` broadcasts them ONCE per new key over RCCL/xGMI
              (a few hundred KB, latency-bound) and every later request with that key is a dictionary hit.
  * exchange  per-utterance results (token ids, audio samples) are variable-length DEVICE tensors keyed by the GLOBAL request
              index (`semantic_orig_idx` semantics kept globally): one all-reduce of a length / owner table, then one
              concatenated buffer per rank -- a padded all-gather (every rank wants them) or point-to-point to one rank
              (`exchange`).  Nothing is pickled; `gather` remains for small host objects only.

Decoding is placement-invariant: rows are independent through every kernel, and device sampling draws a request's noise
from the REQUEST's stream (t2s.py puts request index + 1 into tok_override, gsv_tts_hip.h), not from its slot's.  N ranks
therefore return exactly the tokens one rank returns, greedy or sampled with the same generator seed
(tests/test_hip_engine.py, tests/test_engine_gloo.py).
"""
from __future__ import annotations

import itertools
import os
import threading
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


# GSV_FORCE_COLLECTIVES=1 (or engine.FORCE_COLLECTIVES = True): an initialised process group of ONE rank still goes through the
# backend -- the store cursor, the broadcasts, the all-reduce / all-gather of `exchange` -- instead of the single-process short
# cuts.  A 1-GPU box proves with it that RCCL initialises and runs every collective this path issues (tests/test_hip_rccl.py).
FORCE_COLLECTIVES = os.environ.get("GSV_FORCE_COLLECTIVES") == "1"


def _dist_on(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES)


# ----------------------------------------------------------------------------------------------- dealing
class RequestSource:
    """Hands out global request indices in `order`.  Single process: sequentially.  Multi-rank: `chunk` at a time
    from a cursor shared through `store` (every rank constructs the source with the same `order` and `key`)."""

    def __init__(self, order: Sequence[int], store=None, key: str = "gsv/cursor", chunk: int = 2, world: int = 1):
        self.order = list(order)
        self.store, self.key, self.chunk = store, key, max(1, int(chunk))
        self.world = max(1, int(world))      # ranks pulling from the cursor (fair_share)
        self._local: List[int] = []
        self._pos = 0            # store-less cursor
        self.taken: List[int] = []

    def next(self) -> Optional[int]:
        if not self._local:
            if self.store is None:
                lo, hi = self._pos, min(self._pos + self.chunk, len(self.order))
                self._pos = hi
            else:
                hi = int(self.store.add(self.key, self.chunk))      # atomic fetch-add on the rendezvous store
                lo, hi = hi - self.chunk, min(hi, len(self.order))
            if lo >= len(self.order):
                return None
            self._local = self.order[lo:hi]
        i = self._local.pop(0)
        self.taken.append(i)
        return i

    def fair_share(self) -> int:
        """How many more requests this rank may take AHEAD of its free slots without starving the others: what is left of the
        queue over the ranks that pull from it, rounded up.  A rank that prefills ahead (t2s._infer_batched_ahead) asks before each
        packed prompt pass; without the cap one rank could hold `refill_ahead` requests of the queue's tail while other ranks'
        slots sit empty.  One read of the shared cursor (`add(key, 0)`), once per pass."""
        if self.store is None:
            left = len(self.order) - self._pos
        else:
            left = len(self.order) - int(self.store.add(self.key, 0))
        left = max(0, left) + len(self._local)
        return -(-left // self.world)


_RUN_COUNTER = itertools.count()


def _default_store():
    try:
        from torch.distributed.distributed_c10d import _get_default_store
        return _get_default_store()
    except Exception:
        return None


def lpt_order(costs: Sequence[float]) -> List[int]:
    """longest processing time first; ties by index (every rank computes the same order)"""
    return sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))


# ----------------------------------------------------------------------------------------------- speakers
class SpeakerBook:
    """Per-process cache of reference-speaker tensors, filled by ONE broadcast per new key."""

    def __init__(self, device, group=None):
        self.device, self.group = torch.device(device), group
        self.entries: Dict[str, List[torch.Tensor]] = {}
        self.broadcasts = 0      # number of tensor broadcasts issued (tests / bench read it)

    def sync(self, key: str, tensors: Optional[List[torch.Tensor]], src: int = 0) -> List[torch.Tensor]:
        """`tensors` is the list on `src` (None elsewhere).  Returns the list on every rank."""
        if key in self.entries:
            return self.entries[key]
        if not _dist_on(self.group):
            assert tensors is not None, "single process: the speaker tensors must be given"
            self.entries[key] = [t.to(self.device) for t in tensors]
            return self.entries[key]
        rank = dist.get_rank(self.group)
        meta = [[(tuple(t.shape), str(t.dtype).replace("torch.", "")) for t in tensors]] if rank == src else [None]
        dist.broadcast_object_list(meta, src=src, group=self.group)
        out = []
        for i, (shape, dt) in enumerate(meta[0]):
            t = tensors[i].to(self.device).contiguous() if rank == src else \
                torch.empty(shape, dtype=getattr(torch, dt), device=self.device)
            dist.broadcast(t, src=src, group=self.group)
            self.broadcasts += 1
            out.append(t)
        self.entries[key] = out
        return out


# ----------------------------------------------------------------------------------------------- engine
class ContinuousBatchingEngine:
    """Drives `decoder.infer_batched` (the reference's slot loop) on this rank's share of a global request list.

    decoder   a Text2SemanticDecoder-like object: infer_batched(xs, ys, berts, ..., source=, slots=) ->
              (list of token tensors in completion order, tensor of GLOBAL request indices)
    vocode    optional callable(list of (global index, tokens)) -> dict index -> payload (the vocoder stage of this rank)
    """

    def __init__(self, decoder, slots: int = 32, chunk: int = 2, group=None, store=None):
        self.decoder, self.slots, self.chunk, self.group = decoder, int(slots), int(chunk), group
        self.store = store if store is not None else (_default_store() if _dist_on(group) else None)
        self.rank = dist.get_rank(group) if _dist_on(group) else 0
        self.world = dist.get_world_size(group) if _dist_on(group) else 1
        self.device = getattr(decoder, "device", torch.device("cpu"))
        self.last_taken: List[int] = []
        self._cursor_keys: List[str] = []

    def _source(self, costs: Sequence[float]) -> RequestSource:
        order = lpt_order(costs)
        run = next(_RUN_COUNTER)        # SPMD: every rank makes the same sequence of runs -> the same key
        if self.world == 1 and not (FORCE_COLLECTIVES and self.store is not None):
            return RequestSource(order, None, chunk=len(order) or 1)
        if self.store is None:          # no store: static length-balanced partition (scheduler.shard_indices)
            from .scheduler import shard_indices
            mine = set(shard_indices([int(c) for c in costs], self.world, self.rank))
            return RequestSource([i for i in order if i in mine], None, chunk=len(order) or 1)
        key = "gsv/cursor/%d" % run
        self._cursor_keys.append(key)
        return RequestSource(order, self.store, key=key, chunk=self.chunk, world=self.world)

    def _retire_cursors(self, dst: Optional[int] = None):
        """a finished run's cursor key leaves the rendezvous store (a long-lived server would grow it by one key per
        infer_batched call).  Call it right after a gather / exchange of that run's results: the rank that RECEIVED from
        everybody (`dst`; rank 0 after an all-gather) knows every slot loop has returned, so nobody can add to the key
        any more -- a delete by anyone else could hand the queue out a second time."""
        keys, self._cursor_keys = self._cursor_keys, []
        if self.rank == (0 if dst is None else dst) and self.store is not None:
            for k in keys:
                try:
                    self.store.delete_key(k)
                except Exception:      # a store without delete (FileStore): the key stays, nothing else depends on it
                    pass

    def run_gpt(self, xs, ys, berts, costs: Optional[Sequence[float]] = None, **sampling):
        """-> (pred, idx): this rank's finished requests, completion order, GLOBAL indices"""
        if costs is None:
            costs = [int(x.shape[0]) for x in xs]
        src = self._source(costs)
        pred, idx = self.decoder.infer_batched(xs, ys, berts, source=src, slots=self.slots, **sampling)
        self.last_taken = list(src.taken)
        return pred, idx

    def run_overlapped(self, xs, ys, berts, vocode_batch: Callable, batch: int = 10, costs=None, **sampling):
        """GPT on this rank's share with the vocoder OVERLAPPED: every `batch` finished utterances (the reference's
        sovits_batch_size, TTS.py:728) go to `vocode_batch(list of (index, tokens)) -> dict` on a side stream while the
        slot loop keeps decoding -- the loop is latency-bound on a few CUs per kernel, so the vocoder's wide kernels
        fill the rest of the chip.  (Batches form in completion order here; TTS.infer_batched's length-balanced order,
        TTS.py:705-720, needs all lengths first and is what the non-overlapped path keeps.)
        -> (this rank's results dict index -> payload, pred, idx)"""
        dev = self.decoder.device
        side = torch.cuda.Stream(device=dev)
        pending: List = []
        results: Dict[int, object] = {}

        def flush():
            if not pending:
                return
            items = pending[:]
            pending.clear()
            ev = torch.cuda.Event()
            ev.record()                      # the token tensors were produced on the current stream
            side.wait_event(ev)
            with torch.cuda.stream(side):
                results.update(vocode_batch(items))

        def on_finish(i, tok):
            pending.append((int(i), tok))
            if len(pending) >= batch:
                flush()

        if costs is None:
            costs = [int(x.shape[0]) for x in xs]
        src = self._source(costs)
        pred, idx = self.decoder.infer_batched(xs, ys, berts, source=src, slots=self.slots, on_finish=on_finish, **sampling)
        self.last_taken = list(src.taken)
        flush()
        side.synchronize()
        return results, pred, idx

    def gather(self, local: Dict[int, object], n_total: int, dst: Optional[int] = 0) -> Optional[List[object]]:
        """index -> (small, picklable) payload of this rank  =>  the full list in global index order on rank `dst` (None on
        the other ranks), or on every rank with dst=None.  Tensors go through `exchange`, not through here."""
        if self.world == 1 and not _dist_on(self.group):
            parts = [local]
        elif dst is None:
            parts = [None] * self.world
            dist.all_gather_object(parts, local, group=self.group)
        else:
            parts = [None] * self.world if self.rank == dst else None
            dist.gather_object(local, parts, dst=dst, group=self.group)
            if parts is None:
                return None
        out: List[object] = [None] * n_total
        filled = set()
        for p in parts:
            for i, v in p.items():
                if not 0 <= int(i) < n_total:
                    raise RuntimeError("gather: request index %r outside [0, %d)" % (i, n_total))
                if i in filled:
                    raise RuntimeError("request %d was produced by two ranks" % i)
                filled.add(i)
                out[i] = v
        if len(filled) != n_total:
            raise RuntimeError("gather: %d of %d requests returned" % (len(filled), n_total))
        return out

    def _staged(self) -> bool:
        """gloo moves host tensors only (CI on a 1-GPU box); RCCL moves device tensors over xGMI"""
        return dist.get_backend(self.group) != "nccl"

    def exchange(self, local: Dict[int, torch.Tensor], n_total: int, dst: Optional[int] = 0) -> Optional[List[torch.Tensor]]:
        """Variable-length 1-D tensors keyed by GLOBAL request index (token ids, audio samples), all of one dtype and on
        one device  =>  the list in index order on rank `dst` (None elsewhere), or on every rank with dst=None.
        Device-side: one all-reduce of the [3, n_total] length / owner / count table, then each rank's tensors travel as ONE
        concatenated buffer -- point-to-point to `dst` (RCCL send/recv over the peer's xGMI link: nothing is pickled, nothing
        touches the host), or an all-gather padded to the largest rank when every rank wants them."""
        for i in local:
            if not 0 <= int(i) < n_total:
                raise RuntimeError("exchange: request index %r outside [0, %d)" % (i, n_total))
        if self.world == 1 and not _dist_on(self.group):
            if len(local) != n_total:
                raise RuntimeError("exchange: %d of %d requests present" % (len(local), n_total))
            return [local[i] for i in range(n_total)]
        mine = sorted(local)
        ref = local[mine[0]] if mine else None
        dev = self.device if ref is None else ref.device
        tab = torch.zeros(3, n_total, dtype=torch.int64)
        for i in mine:
            tab[0, i], tab[1, i], tab[2, i] = local[i].numel(), self.rank, 1
        tab = tab.to(dev)
        dist.all_reduce(tab, group=self.group)
        tab = tab.cpu()
        bad = (tab[2] != 1).nonzero().flatten().tolist()
        if bad:
            raise RuntimeError("exchange: requests %s were produced by %s ranks" % (bad[:8], tab[2][bad[:8]].tolist()))
        lens, owner = tab[0].tolist(), tab[1].tolist()
        totals = [0] * self.world
        for l, o in zip(lens, owner):
            totals[o] += l
        dtype = self._exchange_dtype(ref, dev)
        cat = torch.cat([local[i].reshape(-1) for i in mine]) if mine else torch.empty(0, dtype=dtype, device=dev)
        staged = self._staged()
        cdev = torch.device("cpu") if staged else dev
        bufs: List[Optional[torch.Tensor]] = [None] * self.world
        if dst is None:
            cap = max(totals)
            pad = torch.zeros(cap, dtype=dtype, device=cdev)
            pad[:cat.numel()] = cat.to(cdev)
            got = [torch.empty(cap, dtype=dtype, device=cdev) for _ in range(self.world)]
            dist.all_gather(got, pad, group=self.group)
            bufs = [g[:t] for g, t in zip(got, totals)]
        else:
            ops = []
            if self.rank == dst:
                for r in range(self.world):
                    if r == dst:
                        bufs[r] = cat.to(cdev)
                    elif totals[r]:
                        bufs[r] = torch.empty(totals[r], dtype=dtype, device=cdev)
                        ops.append(dist.P2POp(dist.irecv, bufs[r], self._global_rank(r), group=self.group))
                    else:
                        bufs[r] = torch.empty(0, dtype=dtype, device=cdev)
            elif cat.numel():
                ops.append(dist.P2POp(dist.isend, cat.to(cdev).contiguous(), self._global_rank(dst), group=self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            if self.rank != dst:
                return None
        out: List[torch.Tensor] = [None] * n_total
        pos = [0] * self.world
        for i in range(n_total):
            o, l = owner[i], lens[i]
            out[i] = bufs[o][pos[o]:pos[o] + l]
            pos[o] += l
        return [t.to(dev) for t in out] if staged else out

    def _global_rank(self, r: int) -> int:
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def _exchange_dtype(self, ref, dev):
        """every rank must use one dtype, also a rank that holds nothing: agree on it through a tiny all-reduce"""
        codes = [torch.float32, torch.int64, torch.int32, torch.float16, torch.bfloat16, torch.uint8]
        c = torch.tensor([codes.index(ref.dtype) + 1 if ref is not None else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group)
        if int(c.item()) == 0:
            return torch.float32
        return codes[int(c.item()) - 1]

    def deal_batches(self, n_batches: int) -> List[int]:
        """vocoder batches (TTS.py:728-764) are dealt round-robin: batch b runs on rank b mod world.  The batches come out of
        the reference's length-balancing interleave, so their totals are near-equal and a static deal is balanced."""
        return [b for b in range(n_batches) if b % self.world == self.rank]

    def run(self, xs, ys, berts, costs=None, vocode: Optional[Callable] = None, dst: Optional[int] = None, **sampling):
        """GPT on this rank's share, optional per-rank vocoder stage, gather in input order."""
        pred, idx = self.run_gpt(xs, ys, berts, costs, **sampling)
        items = list(zip(idx.tolist(), pred))
        local = vocode(items) if vocode is not None else {int(i): p.cpu() for i, p in items}
        out = self.gather(local, len(xs), dst=dst)
        self._retire_cursors(dst)
        return out
