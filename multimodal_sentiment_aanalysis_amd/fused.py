"""Fused training step on the flat buffers: the body of the reference's `Trainer.train_epoch` loop
(Trainer.py:59-81: zero_grad -> forward -> CE -> backward -> clip_grad_norm_(1.0) -> AdamW.step) with

  * CE forward + dlogits in one kernel (mmsa_ce_fwd_bwd), backward entered with that gradient,
  * gradients written once into the flat fp32 buffer by the engines (overwrite mode: no zero_grad pass),
  * data-parallel: per-engine gradient ranges all-reduced (RCCL = torch.distributed "nccl") on a side HIP stream as
    soon as that engine's backward kernels are enqueued, overlapping with the remaining backward,
  * global-norm clip + AdamW + bf16 working-copy refresh as two kernels over the flat buffers
    (mmsa_grad_norm, mmsa_adamw_step), the 1/world_size average folded into their grad_scale.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, ptr, stream_ptr
from .engine import HeadEngine, engines_of, materialize


class GradReducer:
    """Bucketed SUM all-reduce of gradient ranges, overlapped with the backward that produces them.

    The encoder backwards announce their gradient ranges a few layers / one stage at a time (mmsa_*_bwd_cb: ranges arrive from
    the end of an engine's parameters to their start). `add()` coalesces adjacent announcements until `min_bucket_bytes` are
    pending, then records an event on the producing stream and issues the all-reduce (split at `bucket_bytes`) on a SIDE
    stream that waits for that event only — so the collective of the last layers runs under the backward of the earlier
    ones. `finish()` joins the side stream before the clip. Works on CPU tensors with the gloo backend too (used by the
    world_size-2 CPU tests), where the collectives run inline."""

    def __init__(self, flat_g, bucket_bytes=64 << 20, group=None, min_bucket_bytes=16 << 20, payload="fp32"):
        """payload "bf16" (GPU only): a range is cast to a bf16 staging buffer on the side stream, all-reduced as bf16 (half the
        bytes on xGMI, half of RCCL's active time beside the backward) and widened back into the fp32 gradient buffer; the sum
        over ranks then carries bf16 rounding (2^-9 relative per addition), as DDP's bf16 compression hook does."""
        self.flat_g, self.group = flat_g, group
        self.payload = payload if flat_g.is_cuda else "fp32"
        self._stage = None
        self.bucket_elems = max(1, bucket_bytes // flat_g.element_size())
        self.min_elems = max(1, min_bucket_bytes // flat_g.element_size())
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.works = []
        self.pending = None  # (start, length) announced, not yet issued
        self.issued = []     # (start, length) of every collective of the current step, in issue order (tests)
        self.stream = torch.cuda.Stream(device=flat_g.device) if flat_g.is_cuda else None

    def buckets(self, start, length):
        out, a, end = [], start, start + length
        while a < end:
            b = min(end, a + self.bucket_elems)
            out.append((a, b))
            a = b
        return out

    def add(self, start, length):
        """Announce grad[start, start + length) as enqueued. Adjacent ranges (either side) are merged; a non-adjacent one
        flushes what was pending."""
        if self.world == 1 or length <= 0:
            return
        if self.pending is not None:
            ps, pl = self.pending
            if start + length == ps:
                self.pending = (start, pl + length)
            elif ps + pl == start:
                self.pending = (ps, pl + length)
            else:
                self.flush()
                self.pending = (start, length)
        else:
            self.pending = (start, length)
        if self.pending[1] >= self.min_elems:
            self.flush()

    def flush(self):
        if self.pending is None:
            return
        start, length = self.pending
        self.pending = None
        self.reduce_range(start, length)

    def reduce_range(self, start, length):
        if self.world == 1 or length <= 0:
            return
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat_g.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                for a, b in self.buckets(start, length):
                    if self.payload == "bf16":
                        self._reduce_bf16(a, b)
                    else:
                        self.works.append(dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                                          async_op=True))
                    self.issued.append((a, b - a))
        else:
            for a, b in self.buckets(start, length):
                self.works.append(dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                                  async_op=True))
                self.issued.append((a, b - a))

    def _reduce_bf16(self, a, b):
        """On the side stream: cast -> all-reduce(bf16) -> widen, stream-ordered (the collective is enqueued synchronously with
        respect to this stream, so the widening kernel runs after it). The staging buffer mirrors the gradient buffer."""
        L = _lib.load()
        if self._stage is None:
            self._stage = torch.empty(self.flat_g.numel(), dtype=torch.bfloat16, device=self.flat_g.device)
        sp = ctypes.c_void_p(self.stream.cuda_stream)
        n = b - a
        check(L.mmsa_cast_f32(_lib.MMSA_BF16, ptr(self.flat_g[a:b]), ptr(self._stage[a:b]), n, sp), "mmsa_cast_f32")
        dist.all_reduce(self._stage[a:b], op=dist.ReduceOp.SUM, group=self.group)
        check(L.mmsa_widen_bf16(ptr(self._stage[a:b]), ptr(self.flat_g[a:b]), n, sp), "mmsa_widen_bf16")

    def begin_step(self):
        self.issued = []

    def finish(self):
        self.flush()
        for w in self.works:
            w.wait()
        self.works = []
        if self.stream is not None:
            torch.cuda.current_stream(self.flat_g.device).wait_stream(self.stream)


class ShardedGradReducer(GradReducer):
    """The reduce-scatter form of the data-parallel step ("prefer reduce-scatter + all-gather"): every bucket
    is reduce-SCATTERED instead of all-reduced — rank r ends up with the reduced sum of chunk r of the bucket only — the clip
    norm is a one-double all-reduce of per-rank partial sums of squares, AdamW steps the owned chunks only (1/world of the 30
    B/parameter optimizer pass per GPU), and the updated weights are all-GATHERED back (`gather()`): as fp32 master weights
    (every replica keeps a complete, checkpointable state; same bytes on xGMI as the all-reduce) or as the bf16 working copy the
    encoders compute with (`gather_dtype="bf16"`: half the gather bytes; the fp32 master of chunks a rank does not own then goes
    stale until `sync_master()`).

    A bucket [a, b) is cut into world x chunk elements (chunk a multiple of `align`) + a tail of fewer than world x align elements;
    the tail is all-reduced and stepped by every rank (identical results). The cut depends only on the announced ranges, which a
    configuration repeats step after step: the layout of the first step is recorded and every later step must reproduce it (the
    AdamW moments of an element live on the rank that owns it).

    Backends without reduce_scatter_tensor / all_gather_into_tensor (gloo: the world_size-2 CPU tests and the two-ranks-on-one-GPU
    tests) emulate them with all_reduce / all_gather on the same views, so the bookkeeping is exercised everywhere."""

    def __init__(self, flat_g, bucket_bytes=64 << 20, group=None, min_bucket_bytes=16 << 20, payload="fp32", align=256):
        super().__init__(flat_g, bucket_bytes, group, min_bucket_bytes, payload)
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.align = int(align)
        self.native = self.world > 1 and dist.get_backend(group) == "nccl"
        self.owned, self.tails, self.mains = [], [], []
        self._layout = None

    def begin_step(self):
        super().begin_step()
        self.owned, self.tails, self.mains = [], [], []

    def _cut(self, a, b):
        n = b - a
        chunk = (n // (self.world * self.align)) * self.align
        return chunk, chunk * self.world

    def reduce_range(self, start, length):
        if self.world == 1 or length <= 0:
            return
        ctx = None
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat_g.device))
            ctx = torch.cuda.stream(self.stream)
            ctx.__enter__()
            self.stream.wait_event(ev)
        try:
            for a, b in self.buckets(start, length):
                chunk, main = self._cut(a, b)
                if chunk > 0:
                    self._reduce_scatter(a, chunk, main)
                    self.owned.append((a + self.rank * chunk, chunk))
                    self.mains.append((a, main, chunk))
                if main < b - a:  # the remainder: all-reduced, stepped by every rank
                    self.works.append(dist.all_reduce(self.flat_g[a + main:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    self.tails.append((a + main, b - a - main))
                self.issued.append((a, b - a))
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def _reduce_scatter(self, a, chunk, main):
        view = self.flat_g[a:a + main]
        own = view[self.rank * chunk:(self.rank + 1) * chunk]
        if self.payload == "bf16" and self.stream is not None:
            L = _lib.load()
            if self._stage is None:
                self._stage = torch.empty(self.flat_g.numel(), dtype=torch.bfloat16, device=self.flat_g.device)
            sp = ctypes.c_void_p(self.stream.cuda_stream)
            sv = self._stage[a:a + main]
            so = sv[self.rank * chunk:(self.rank + 1) * chunk]
            check(L.mmsa_cast_f32(_lib.MMSA_BF16, ptr(view), ptr(sv), main, sp), "mmsa_cast_f32")
            if self.native:
                dist.reduce_scatter_tensor(so, sv, op=dist.ReduceOp.SUM, group=self.group)
            else:
                dist.all_reduce(sv, op=dist.ReduceOp.SUM, group=self.group)
            check(L.mmsa_widen_bf16(ptr(so), ptr(own), chunk, sp), "mmsa_widen_bf16")
        elif self.native:
            self.works.append(dist.reduce_scatter_tensor(own, view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:  # emulation: the whole body is reduced; only the owned chunk is used afterwards
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        super().finish()
        layout = (tuple(self.mains), tuple(self.tails))
        if self._layout is None:
            self._layout = layout
        elif layout != self._layout:
            raise _lib.MmsaError("ShardedGradReducer: the bucket layout changed between steps (the AdamW moments of an element live "
                                 "on its owner): rebuild the step after changing which parameters train")

    def step_ranges(self):
        """Ranges this rank's optimizer steps: its chunks and the replicated tails."""
        return merge_ranges(self.owned + self.tails)

    def norm_ranges(self):
        """Ranges whose squares this rank contributes to the global norm: its chunks; the tails are counted once, by rank 0."""
        return merge_ranges(self.owned + (self.tails if self.rank == 0 else []))

    def gather(self, state, gather_dtype="fp32"):
        """All-gather the stepped chunks back into every replica (after the optimizer). fp32: the master weights, then a local
        re-cast of the bf16 working copy of the chunks other ranks stepped. bf16: the working copy itself where a bucket lies wholly
        inside bf16 engines (the fp32 master of foreign chunks goes stale: sync_master()); fp32 elsewhere (the fusion head computes
        in fp32 on the master weights)."""
        if self.world == 1:
            return
        L = _lib.load()
        bf16_spans = merge_ranges([(off, n) for e, off, n in state.ranges if getattr(e, "precision", "fp32") in ("bf16", "fp8")])
        for a, main, chunk in self.mains:
            lo = a + self.rank * chunk
            inside = state.flat_wt is not None and any(s0 <= a and a + main <= s0 + n0 for s0, n0 in bf16_spans)
            if gather_dtype == "bf16" and inside:
                self._all_gather(state.flat_wt[a:a + main], chunk)
                continue
            self._all_gather(state.flat_w[a:a + main], chunk)
            if state.flat_wt is not None and state.flat_w.is_cuda:
                for s0, n0 in ((a, lo - a), (lo + chunk, a + main - lo - chunk)):
                    if n0 > 0:
                        check(L.mmsa_cast_f32(_lib.MMSA_BF16, ptr(state.flat_w[s0:s0 + n0]), ptr(state.flat_wt[s0:s0 + n0]), n0,
                                              stream_ptr()), "mmsa_cast_f32")

    def sync_master(self, state):
        """bf16 gathers only: bring the fp32 master weights of every replica up to date (before a checkpoint / an evaluation that
        reads `state_dict()`)."""
        for a, main, chunk in self.mains:
            self._all_gather(state.flat_w[a:a + main], chunk)

    def _all_gather(self, view, chunk):
        own = view[self.rank * chunk:(self.rank + 1) * chunk]
        if self.native:
            dist.all_gather_into_tensor(view, own, group=self.group)
        else:
            outs = [view[r * chunk:(r + 1) * chunk] for r in range(self.world)]
            dist.all_gather(outs, own.clone(), group=self.group)


class FlatAdamW:
    """clip_grad_norm_(max_norm) + AdamW over a FlatState's buffers (Trainer.py:19-21,80-81) on the HIP kernels.

    `ranges`: [(offset, length)] in elements of the flat buffers — the parameters this optimizer owns (default: all of
    them). A curriculum phase (dataLoader/MultiTaskTrainer.py) hands over the ranges of its trainable modules: the norm
    is taken over exactly those, the update touches nothing else. The decision "is this step applied?" is made on the
    device (Trainer.py:74-76 skips the update on a NaN loss): a non-finite gradient norm or loss leaves w, m, v and the bf16
    working copy untouched and does not advance the bias-correction step count (a device int32)."""

    def __init__(self, state, lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0, ranges=None,
                 norm_ranges=None, clip_in_place=False):
        """clip_in_place: also scale the optimizer's OWN gradients by the clip coefficient after the step, as
        clip_grad_norm_ does (the AdamW kernel applies the coefficient on the fly and leaves g alone; only a caller that reads or
        re-uses those gradients after the step — a later curriculum phase whose clip norm includes them — needs them scaled)."""
        self.clip_in_place = bool(clip_in_place)
        self.state, self.lr, self.wd, self.betas, self.eps, self.max_norm = state, lr, weight_decay, betas, eps, max_norm
        dev = state.flat_w.device
        n = state.flat_w.numel()
        self.ranges = merge_ranges([(0, n)] if ranges is None else ranges)
        # the clip norm may span more than the optimizer owns (clip_grad_norm_ over every trainable parameter while the
        # optimizer steps a subset: MultiTaskTrainer.py:147-177); the extra ranges are scaled in place like torch does
        self.norm_ranges = self.ranges if norm_ranges is None else merge_ranges(list(norm_ranges) + self.ranges)
        self.extra_ranges = subtract_ranges(self.norm_ranges, self.ranges)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.steps = torch.zeros(1, dtype=torch.int32, device=dev)  # applied (not skipped) steps
        self.norm_ws = torch.empty(_lib.load().mmsa_grad_norm_ws_bytes(), dtype=torch.uint8, device=dev)
        # [total norm, clip coefficient | -1 = skipped, 1 - beta1^t, sqrt(1 - beta2^t)] (t = applied steps, this one included)
        self.norm_out = torch.zeros(4, dtype=torch.float32, device=dev)
        self._set_norm_arrays()

    def _set_norm_arrays(self):
        if len(self.norm_ranges) == 0:  # (a rank of the reduce-scatter step may own nothing of a tiny model)
            self._offs, self._lens = (ctypes.c_int64 * 1)(0), (ctypes.c_int64 * 1)(0)
            return
        if len(self.norm_ranges) > 256:
            raise _lib.MmsaError(f"{len(self.norm_ranges)} disjoint gradient ranges in one clip norm (limit 256)")
        self._offs = (ctypes.c_int64 * len(self.norm_ranges))(*[a for a, _ in self.norm_ranges])
        self._lens = (ctypes.c_int64 * len(self.norm_ranges))(*[n_ for _, n_ in self.norm_ranges])

    def set_ranges(self, ranges, norm_ranges):
        """Replace the ranges stepped and the ranges the clip norm spans (made a superset of the former): the set of parameters
        that hold a gradient can change from step to step (PhaseOptimizer). m and v span the whole buffer, so nothing moves."""
        self.ranges = merge_ranges(ranges)
        self.norm_ranges = merge_ranges(list(norm_ranges) + self.ranges)
        self.extra_ranges = subtract_ranges(self.norm_ranges, self.ranges)
        self._set_norm_arrays()

    def set_shard_ranges(self, step_ranges, norm_ranges):
        """The reduce-scatter step: the ranges this rank steps (its gradient shards + the replicated tails) and the ranges whose
        squares it contributes to the global norm (its shards; the tails on rank 0 only) are set independently."""
        self.ranges = merge_ranges(step_ranges)
        self.norm_ranges = merge_ranges(norm_ranges)
        self.extra_ranges = []
        self._set_norm_arrays()

    @property
    def t(self):
        """Number of applied steps (host read: a sync; for tests and checkpoints only)."""
        return int(self.steps.item())

    def step(self, grad_scale=1.0, loss=None, sumsq_group=None):
        """sumsq_group (the reduce-scatter step): this rank's norm ranges cover only the gradient shards it owns; the partial sums of
        squares of all ranks are added by a one-double all-reduce over that process group before the norm is finalized."""
        L = _lib.load()
        st = self.state
        if sumsq_group is not None:
            if getattr(self, "_sumsq", None) is None:
                self._sumsq = torch.zeros(1, dtype=torch.float64, device=st.flat_g.device)
            check(L.mmsa_grad_sumsq_ranges(ptr(st.flat_g), self._offs, self._lens, len(self.norm_ranges), ptr(self._sumsq),
                                           ptr(self.norm_ws), stream_ptr()), "mmsa_grad_sumsq_ranges")
            dist.all_reduce(self._sumsq, op=dist.ReduceOp.SUM, group=None if sumsq_group is True else sumsq_group)
            check(L.mmsa_grad_norm_from_sumsq(ptr(self._sumsq), 1, grad_scale, self.max_norm, ptr(loss), ptr(self.steps),
                                              ptr(self.norm_out), self.betas[0], self.betas[1], stream_ptr()),
                  "mmsa_grad_norm_from_sumsq")
        else:
            check(L.mmsa_grad_norm_ranges(ptr(st.flat_g), self._offs, self._lens, len(self.norm_ranges), grad_scale, self.max_norm,
                                          ptr(loss), ptr(self.steps), ptr(self.norm_out), ptr(self.norm_ws), self.betas[0],
                                          self.betas[1], stream_ptr()),
                  "mmsa_grad_norm_ranges")
        for a, n in self.ranges:
            w16 = None if st.flat_wt is None else st.flat_wt[a:a + n]
            check(L.mmsa_adamw_step_dev(ptr(st.flat_w[a:a + n]), ptr(st.flat_g[a:a + n]), ptr(self.m[a:a + n]),
                                        ptr(self.v[a:a + n]), ptr(w16), n, self.lr, self.betas[0], self.betas[1], self.eps,
                                        self.wd, ptr(self.steps), ptr(self.norm_out), grad_scale, stream_ptr()),
                  "mmsa_adamw_step_dev")
        for a, n in (self.extra_ranges + self.ranges if self.clip_in_place else self.extra_ranges):
            check(L.mmsa_grad_scale_clip(ptr(st.flat_g[a:a + n]), n, ptr(self.norm_out), stream_ptr()), "mmsa_grad_scale_clip")
        if self.ranges == [(0, st.flat_w.numel())]:
            for e, _, _ in st.ranges:
                if not isinstance(e, HeadEngine):
                    e.mark_weights_fresh()  # the whole bf16 working copy was rewritten by this step
        # (a partial step needs nothing: the kernel refreshed the bf16 working copy of exactly the ranges it stepped, the rest
        #  of the master did not change, and the engines' version tokens still describe it)


def subtract_ranges(a, b):
    """Elements of the (merged) ranges `a` that are in none of the (merged) ranges `b`, as merged ranges."""
    out = []
    for s0, n0 in a:
        cur = s0
        for s1, n1 in b:
            if s1 + n1 <= cur or s1 >= s0 + n0:
                continue
            if s1 > cur:
                out.append((cur, s1 - cur))
            cur = max(cur, s1 + n1)
        if cur < s0 + n0:
            out.append((cur, s0 + n0 - cur))
    return out


def param_ranges(state, params):
    """Flat-buffer ranges [(offset, length)] of the given parameters (those that live in the flat buffers), padded to the
    64-element alignment the tables use (the padding holds zeros in w, g, m and v and stays zero under AdamW), merged."""
    base = state.flat_w.data_ptr()
    total = state.flat_w.numel()
    out = []
    seen = set()
    for p in params:
        if id(p) in seen:
            continue
        seen.add(id(p))
        off = (p.data_ptr() - base) // 4
        if 0 <= off < total:
            out.append((off, min((p.numel() + 63) // 64 * 64, total - off)))
    return merge_ranges(out)


def module_ranges(state, modules):
    """param_ranges of every parameter of `modules`."""
    return param_ranges(state, [p for m in modules for p in m.parameters()])


class PhaseOptimizer:
    """What a curriculum phase of the reference's MultiTaskTrainer builds with torch (`optim.AdamW(params, lr=1e-4,
    weight_decay=1e-4)` + `clip_grad_norm_(self.model.parameters(), 1.0)`, MultiTaskTrainer.py:55-177,179-467), on the HIP
    kernels over sub-ranges of the flat buffers: `step()` = clip + AdamW on `opt_modules`.

    The clip is the reference's: over EVERY parameter of the model whose `.grad` is not None at that moment
    (`clip_grad_norm_(self.model.parameters(), ...)`, MultiTaskTrainer.py:205,261,317,378,439) — the trainable modules of the
    phase AND anything frozen now that still holds a gradient from an earlier phase (arousal_head and the encoders while
    phase 3 runs, eeg_net while the eye phase runs): those stale gradients enter every norm and are rescaled in place by every
    clip. `zero_grad()` drops the optimizer's own gradients only (torch's set_to_none: the next backward re-creates them from
    zero) — the reference's phase 3 never zeroes the modules it unfreezes but does not optimize, so they keep accumulating.
    `param_groups[0]["lr"]` is live, so a plateau scheduler can drive it. `norm_history` (when a list) receives a clone of
    the device-side [norm, clip coefficient, ...] of every step (tests compare it with the reference's return values)."""

    def __init__(self, state, opt_modules, model, lr=1e-4, weight_decay=1e-4, max_norm=1.0):
        self.state, self.model = state, model
        self.opt_params = [p for m in opt_modules for p in m.parameters()]
        self.opt_ranges = param_ranges(state, self.opt_params)
        self.param_groups = [{"lr": lr}]
        # clip_in_place: the gradients a phase leaves behind are the CLIPPED ones (torch scales .grad in place), and the next
        # phase's norms include them (stale gradients of modules frozen later; accumulation onto them in phase 3)
        self.adamw = FlatAdamW(state, lr=lr, weight_decay=weight_decay, max_norm=max_norm, ranges=self.opt_ranges,
                               clip_in_place=True)
        self.norm_history = None
        self._live_key = None

    def zero_grad(self):
        for p in self.opt_params:
            p.grad = None

    def live_ranges(self):
        """Ranges of every parameter that holds a gradient now (`p.grad is not None`): what clip_grad_norm_ sees; the
        optimizer's own parameters among them are what AdamW steps (torch skips a parameter whose .grad is None)."""
        live = [p for p in self.model.parameters() if p.grad is not None]
        key = tuple(id(p) for p in live)
        if key != self._live_key:
            self._live_key = key
            ids = set(key)
            self.adamw.set_ranges(param_ranges(self.state, [p for p in self.opt_params if id(p) in ids]),
                                  param_ranges(self.state, live))
        return self.adamw.norm_ranges

    def step(self):
        self.adamw.lr = self.param_groups[0]["lr"]
        self.live_ranges()
        if not self.adamw.norm_ranges:
            return
        self.adamw.step()
        if self.norm_history is not None:
            self.norm_history.append(self.adamw.norm_out.clone())


def _enclosing_state(module):
    """A valid FlatState (of some larger root) whose buffers already hold every engine of `module`, or None."""
    engs = engines_of(module)
    if not engs or any(e._flat_w is None for e in engs):
        return None
    st = getattr(engs[0], "_owner_state", None)
    st = st() if callable(st) else st
    if st is None or not st.valid():
        return None
    members = {id(e) for e, _, _ in st.ranges}
    return st if all(id(e) in members for e in engs) else None


class FlatAdam:
    """`optim.Adam(list(encoder.parameters()) + list(projection_head.parameters()), lr=lr)` / `optim.Adam(classifier.parameters(),
    lr=lr)` of the reference's two-stage pipeline (train.py:52,94) on the HIP optimizer kernel: Adam is AdamW with zero decay, and
    without a clip the coefficient is 1 (max_norm = inf). One FlatAdamW per flat parameter state (modules that were materialized
    separately — encoder, projection head, classifier — own separate flat buffers); `zero_grad()` is torch's set_to_none."""

    def __init__(self, modules, lr=1e-3, device=None, betas=(0.9, 0.999), eps=1e-8):
        self.param_groups = [{"lr": lr}]
        self.parts = []
        for m in modules:
            st = getattr(m, "_flat_state", None)
            if st is None or not st.valid():
                # a sub-module of an already materialized model: step the enclosing state's ranges instead of re-binding its
                # engines into buffers of their own (which would invalidate the enclosing FlatState)
                st = _enclosing_state(m)
            if st is None:
                dev = torch.device(device) if device is not None else next(m.parameters()).device
                st = materialize(m, dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device()))
            params = list(m.parameters())
            ranges = param_ranges(st, params)
            if sum(n for _, n in ranges) < sum(p.numel() for p in params):
                raise _lib.MmsaError("FlatAdam: a parameter lives outside the module's flat buffers")
            self.parts.append((params, FlatAdamW(st, lr, 0.0, betas, eps, max_norm=float("inf"), ranges=ranges)))

    def zero_grad(self):
        for params, _ in self.parts:
            for p in params:
                p.grad = None

    def step(self):
        for params, opt in self.parts:
            if any(p.grad is not None for p in params):  # (torch.optim.Adam skips parameters without a gradient)
                opt.lr = self.param_groups[0]["lr"]
                opt.step()


class Plateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode="min", threshold=1e-4 relative, cooldown=0, min_lr=0, eps=1e-8) for
    an optimizer facade with `param_groups` (MultiTaskTrainer.py:70-75,146-151,172-177)."""

    def __init__(self, optimizer, patience, factor):
        self.optimizer, self.patience, self.factor = optimizer, patience, factor
        self.best, self.bad = float("inf"), 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - 1e-4):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            for g in self.optimizer.param_groups:
                new = g["lr"] * self.factor
                if g["lr"] - new > 1e-8:
                    g["lr"] = new
            self.bad = 0


def merge_ranges(ranges):
    """Sort and coalesce [(offset, length)] (touching ranges become one: fewer launches, larger collectives)."""
    out = []
    for a, n in sorted((int(a), int(n)) for a, n in ranges if n > 0):
        if out and out[-1][0] + out[-1][1] >= a:
            end = max(out[-1][0] + out[-1][1], a + n)
            out[-1] = (out[-1][0], end - out[-1][0])
        else:
            out.append((a, n))
    return out


class FusedTrainStep:
    """model: MultimodalTransformerModel (Trainer contract). One call = one optimizer step."""

    def __init__(self, model, device, precision="bf16", lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8,
                 max_norm=1.0, bucket_bytes=64 << 20, two_streams=None, min_bucket_bytes=16 << 20, train_mode=True,
                 shard_optimizer=None, gather_dtype=None):
        """train_mode=False: the forward runs in eval mode (BatchNorm on its running statistics, no dropout) while gradients
        and the optimizer step are still taken — the configuration SURVEY.md section 8(e) prescribes for checking that N ranks x
        B/N samples reproduce one rank x B samples (batch statistics would differ between the two by construction)."""
        self.train_mode = bool(train_mode)
        self.model, self.device = model, torch.device(device)
        self.state = materialize(model, self.device, precision)
        # the optimizer owns the TRAINABLE parameters (Trainer.py:19-21 builds AdamW over model.parameters(); a frozen
        # encoder — train.py:90-92 — must not even decay): ranges of every requires_grad parameter; all of them = one range
        params = list(model.parameters())
        trainable = [p for p in params if p.requires_grad]
        ranges = None if len(trainable) == len(params) else param_ranges(self.state, trainable)
        self.opt = FlatAdamW(self.state, lr, weight_decay, betas, eps, max_norm, ranges=ranges)
        self.loss = torch.zeros((), dtype=torch.float32, device=self.device)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # (No CU reservation for the collective by default. A workgroup of the persistent GEMM fills its CU, so RCCL's blocks
        # wait for whole CUs; MMSA_G2_CUS=<n> caps the GEMM grid to leave some free — but the planner's tile counts are exact
        # multiples of 256 CUs (BERT-base: 768 tiles = 3 rounds), and measured on one MI355X a cap of 248 / 240 / 224 costs
        # 5.3 % / 5.0 % / 12 % of the WHOLE step, forward included (tools/exp_streams.sh). Left to the env.)
        # gradient payload of the all-reduce: MMSA_GRAD_PAYLOAD=fp32 | bf16 (default fp32: bit-identical sums on every rank in
        # rank order independent precision; bf16 halves the bytes on xGMI — DESIGN.md §6)
        payload = os.environ.get("MMSA_GRAD_PAYLOAD", "fp32")
        # shard_optimizer (MMSA_SHARD_OPT=1): the reduce-scatter form of the step (ShardedGradReducer): reduce-scatter of every
        # bucket, one-double all-reduce for the clip norm, AdamW on the owned 1/world of the parameters, all-gather of the updated
        # weights (gather_dtype / MMSA_GATHER_DTYPE: "fp32" master weights — every replica stays checkpointable — or "bf16": the
        # encoders' working copy, half the gather bytes, fp32 master of foreign shards stale until sync_master()). Needs every
        # parameter trainable (the shard layout is cut from the announced ranges).
        if shard_optimizer is None:
            shard_optimizer = os.environ.get("MMSA_SHARD_OPT", "0") == "1"
        self.shard_optimizer = bool(shard_optimizer) and self.world > 1
        self.gather_dtype = gather_dtype or os.environ.get("MMSA_GATHER_DTYPE", "fp32")
        if self.shard_optimizer and ranges is not None:
            raise _lib.MmsaError("shard_optimizer needs every parameter trainable (frozen sub-graphs: use the all-reduce step)")
        rcls = ShardedGradReducer if self.shard_optimizer else GradReducer
        self.reducer = (rcls(self.state.flat_g, bucket_bytes, payload=payload, min_bucket_bytes=min_bucket_bytes)
                        if self.world > 1 else None)
        self.exposed_events = None  # bench.py: [(before, after)] timing events around the join with the collective stream
        if self.world > 1:  # identical replicas: parameters and BN buffers from rank 0
            dist.broadcast(self.state.flat_w, 0)
            dist.broadcast(self.state.flat_bn, 0)
        # The image encoder runs on its own HIP stream beside the text encoder (joined before the head / the optimizer): its
        # many latency-bound kernels (BatchNorm finalizes, small-grid BatchNorm passes, slab reducers) fill the launch bubbles
        # and tails of the text encoder's GEMMs and vice versa. Measured on MI355X, same box: 17.59-17.61 against 17.95-18.03
        # ms/step (round 3). (In rounds 1-2 it was a loss, 19.50 vs 19.23: two persistent GEMMs that meet only stretch each
        # other — the sum of the GEMM kernels' own durations still grows, 10.28 -> 10.98 ms — but the kernels around them got
        # leaner since.) two_streams=None: on unless MMSA_TWO_STREAMS=0.
        if two_streams is None:
            two_streams = os.environ.get("MMSA_TWO_STREAMS", "1") != "0"
        self.two_streams = bool(two_streams)
        self._image_net = getattr(getattr(model, "encoder", None), "image_net", None)
        if self._image_net is not None and self.device.type == "cuda" and two_streams:
            self._image_net.use_side_stream(True)
        # ... and its stage-wise weight-gradient groups on one more stream (mmsa_resnet_bwd_cb2): 17.15 against 17.37-17.46
        # ms/step on the same box; MMSA_WGRAD_STREAM=0 turns it off; ignored by the engine under data parallelism
        if self._image_net is not None and self.device.type == "cuda" and os.environ.get("MMSA_WGRAD_STREAM", "1") != "0":
            self._image_net.use_wgrad_stream(True)
        self._ranges = {id(e): (off, n) for e, off, n in self.state.ranges}
        # (Round 4 measured taking each range's sum of squares the moment the backward announces it — 16 small launches under the
        #  backward instead of one 540 MB pass behind it: 17.04 against 16.82 ms per step; the extra launches stretch the GEMMs they
        #  run beside by more than the 0.09 ms they take out of the tail. Removed.)
        for e, off, n in self.state.ranges:
            on = self.reducer is not None
            e._grad_ready_hook = self._on_grads_ready if on else None
            e._grad_range_hook = self._on_range_ready if on and not isinstance(e, HeadEngine) else None

    def _on_range_ready(self, eng, off, length):  # from inside the encoder's backward: a few layers / a stage at a time
        base, _n = self._ranges[id(eng)]
        self.reducer.add(base + off, length)

    def _on_grads_ready(self, eng):  # end of an engine's backward
        if isinstance(eng, HeadEngine):
            off, n = self._ranges[id(eng)]
            self.reducer.add(off, n)
        else:
            self.reducer.flush()  # the encoder announced everything itself; push out what is still pending

    def step(self, image, token_ids, attention_mask, labels):
        if not self.state.valid():
            raise _lib.MmsaError("flat parameter buffers were invalidated (model.to()/.float() after FusedTrainStep)")
        L = _lib.load()
        model = self.model
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # Stream capture of the step: only the single-stream form is capturable. With the side streams the step creates
            # torch events, hands caching-allocator blocks to streams that were not forked from the capturing one
            # (record_stream) and lets the encoder callbacks record on them — round 3's capture attempt crashed there. Refuse
            # loudly instead (tools/microbench/graph_probe.py captures the single-stream step; a graph buys nothing here: the step
            # is GPU-bound, DESIGN.md section 3 "Host side").
            side = getattr(self._image_net, "_side", None) is not None or getattr(self._image_net, "_wgrad_stream", None) is not None
            if side or self.reducer is not None:
                raise _lib.MmsaError("FusedTrainStep.step cannot be stream-captured with its side streams on: build it with "
                                     "two_streams=False and MMSA_WGRAD_STREAM=0 (single process) to capture the step")
        model.train(self.train_mode)
        if self.reducer is not None:
            self.reducer.begin_step()
        for e, _, _ in self.state.ranges:
            e._overwrite_next = True  # every gradient range is written (not accumulated) by this backward
        logits, _aux = model(image, token_ids, attention_mask, labels)
        B, C = logits.shape
        dlogits = torch.empty_like(logits)
        check(L.mmsa_ce_fwd_bwd(ptr(logits), ptr(labels), ptr(self.loss), ptr(dlogits), None, B, C, 1.0, stream_ptr()),
              "mmsa_ce_fwd_bwd")
        logits.backward(dlogits)
        if self._image_net is not None:
            self._image_net.join()
        if self.reducer is not None:
            if self.exposed_events is not None:  # how long the step waits for the collectives the backward did not hide
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.reducer.finish()
                e1.record()
                self.exposed_events.append((e0, e1))
            else:
                self.reducer.finish()
        # NaN rule (Trainer.py:74-76): decided on the device from the reduced gradient norm (identical on every rank, so
        # the replicas stay in step); a NaN loss on any rank makes its gradients, hence the reduced norm, non-finite
        if self.shard_optimizer:
            self.opt.set_shard_ranges(self.reducer.step_ranges(), self.reducer.norm_ranges())
            self.opt.step(1.0 / self.world, None, sumsq_group=True)
            self.reducer.gather(self.state, self.gather_dtype)
        else:
            self.opt.step(1.0 / self.world, self.loss if self.world == 1 else None)
        return self.loss, logits

    def sync_master(self):
        """After steps with gather_dtype="bf16": all-gather the fp32 master weights (before state_dict() / a checkpoint)."""
        if self.shard_optimizer and self.gather_dtype == "bf16":
            self.reducer.sync_master(self.state)

    @property
    def lr(self):
        return self.opt.lr

    @lr.setter
    def lr(self, v):
        self.opt.lr = v
