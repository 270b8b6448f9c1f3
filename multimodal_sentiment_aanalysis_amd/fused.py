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

    @property
    def t(self):
        """Number of applied steps (host read: a sync; for tests and checkpoints only)."""
        return int(self.steps.item())

    def step(self, grad_scale=1.0, loss=None):
        L = _lib.load()
        st = self.state
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
                 max_norm=1.0, bucket_bytes=64 << 20, two_streams=None, min_bucket_bytes=16 << 20, train_mode=True):
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
        self.reducer = (GradReducer(self.state.flat_g, bucket_bytes, payload=payload, min_bucket_bytes=min_bucket_bytes)
                        if self.world > 1 else None)
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
        # optional: the two cross-modal transformers on side streams beside the fusion chain (MMSA_HEAD_STREAMS=1; A/B hook)
        self._head_streams = None
        if os.environ.get("MMSA_HEAD_STREAMS", "0") == "1" and hasattr(model, "use_head_streams") and self.device.type == "cuda":
            model.use_head_streams(True)
            self._head_streams = model._head_streams
        self._ranges = {id(e): (off, n) for e, off, n in self.state.ranges}
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
            if side or self._head_streams is not None or self.reducer is not None:
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
        if self._head_streams is not None:  # their backward kernels wrote parameter gradients autograd does not track
            for st in self._head_streams:
                torch.cuda.current_stream(self.device).wait_stream(st)
        if self.reducer is not None:
            self.reducer.finish()
        # NaN rule (Trainer.py:74-76): decided on the device from the reduced gradient norm (identical on every rank, so
        # the replicas stay in step); a NaN loss on any rank makes its gradients, hence the reduced norm, non-finite
        self.opt.step(1.0 / self.world, self.loss if self.world == 1 else None)
        return self.loss, logits

    @property
    def lr(self):
        return self.opt.lr

    @lr.setter
    def lr(self, v):
        self.opt.lr = v
