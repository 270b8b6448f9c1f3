"""Engine modules: nn.Modules whose parameters live in flat device buffers laid out by libmmsa_hip.so and whose
forward / backward are single calls into its engines (BERT text encoder, ResNet image encoder, fusion-head modules).

Plumbing only: PyTorch owns the memory, the stream and the autograd graph edges between modules; every FLOP
is executed by the HIP library. There is no CPU path: calling a module on a CPU tensor raises MmsaError.

Parameter storage
  * each engine reports its flat layout (name, offset, shape); parameters are created on the CPU with the usual
    initialisers and keep torch-compatible names/shapes (`state_dict()` interchanges with HF BertModel /
    torchvision ResNet / the reference's fusion modules);
  * on first use on a GPU, `materialize(root)` packs every engine under `root` into ONE fp32 buffer (+ one fp32
    gradient buffer, + a bf16 working copy), and re-points `p.data` / `p.grad` at views of them. Backward writes
    gradients straight into the flat buffer (no autograd accumulation pass), which is also what the fused
    optimizer and the data-parallel reducer operate on.
"""
import contextlib
import ctypes
import math

import torch
import torch.nn as nn

from . import _lib
from ._lib import (HEAD_CLASSIFIER, HEAD_CROSS_MODAL, HEAD_MM_FUSION, HEAD_PROJECTION, HEAD_WEIGHTED, MMSA_BF16,
                   MMSA_F32, MMSA_FP8, RANGE_CB, BertCfg, HeadCfg, MmsaError, ResnetCfg, check, param_table, ptr, ptr_array,
                   stream_ptr)

BERT_BASE = dict(hidden=768, layers=12, heads=12, intermediate=3072, vocab=30522, max_pos=512, type_vocab=2,
                 ln_eps=1e-12)
BERT_LARGE = dict(hidden=1024, layers=24, heads=16, intermediate=4096, vocab=30522, max_pos=512, type_vocab=2,
                  ln_eps=1e-12)
RESNET50 = dict(blocks=(3, 4, 6, 3), widths=(64, 128, 256, 512))
RESNET101 = dict(blocks=(3, 4, 23, 3), widths=(64, 128, 256, 512))


class _Node(nn.Module):
    """Name-space container so flat-table names like `bert.encoder.layer.0.output.dense.weight` become attributes."""


def _attach(root, dotted, tensor, buffer=False):
    parts = dotted.split(".")
    mod = root
    for part in parts[:-1]:
        if part not in mod._modules:
            mod.add_module(part, _Node())
        mod = mod._modules[part]
    if buffer:
        mod.register_buffer(parts[-1], tensor)
        return tensor
    par = nn.Parameter(tensor)
    mod.register_parameter(parts[-1], par)
    return par


def _require_gpu(t, what):
    if not t.is_cuda:
        raise MmsaError(f"{what}: input is on {t.device}; this path runs only on a GPU through libmmsa_hip.so "
                        "(no CPU fallback)")


class EngineModule(nn.Module):
    """Base: flat parameter table + (re)binding into shared flat buffers."""

    # storage type of encoder activations / working weights: "bf16" | "fp32" (heads are always fp32), or "fp8" = bf16 storage with
    # fp8 (e4m3) operands in the text encoder's forward Linears (BASELINE.json configs[4]; the image encoder then runs as bf16)
    precision = "bf16"

    def _setup_tables(self, specs, bspecs, numel, bn_numel):
        self._specs, self._bspecs, self._numel, self._bn_numel = specs, bspecs, int(numel), int(bn_numel)
        self._pmap, self._bmap, self._nbt = {}, {}, {}
        self._flat_w = self._flat_g = self._flat_wt = self._flat_bn = self._flat_nbt = None
        self._dummy = None
        self._ws_pool = []
        self._wt_token = None
        self._overwrite_next = False
        for name, off, shape in specs:
            self._pmap[name] = _attach(self, name, torch.empty(shape, dtype=torch.float32))
        for name, off, shape in bspecs:
            fill = 1.0 if name.endswith("running_var") else 0.0
            self._bmap[name] = _attach(self, name, torch.full(shape, fill, dtype=torch.float32), buffer=True)
            if name.endswith("running_mean"):
                nb = name[:-len("running_mean")] + "num_batches_tracked"
                self._nbt[nb] = _attach(self, nb, torch.zeros((), dtype=torch.long), buffer=True)

    # ---- views ------------------------------------------------------------------------------------------------
    @staticmethod
    def _view(flat, off, shape):
        n = math.prod(shape)
        v = flat[off:off + n]
        if len(shape) == 4:  # conv weight: logical [O,I,KH,KW], physical [O,KH,KW,I] (channels_last)
            o, i, kh, kw = shape
            return v.view(o, kh, kw, i).permute(0, 3, 1, 2)
        return v.view(shape)

    def bind(self, flat_w, flat_g, flat_wt, flat_bn, flat_nbt):
        """Adopt slices of shared flat buffers: copy current values in, re-point parameters, buffers and grads."""
        with torch.no_grad():
            for name, off, shape in self._specs:
                p = self._pmap[name]
                view = self._view(flat_w, off, shape)
                view.copy_(p.data)
                p.data = view
                p.grad = None
            for name, off, shape in self._bspecs:
                b = self._bmap[name]
                view = flat_bn[off:off + math.prod(shape)].view(shape)
                view.copy_(b)
                self._rebuffer(name, view)
            for k, (name, old) in enumerate(list(self._nbt.items())):
                view = flat_nbt[k]
                view.copy_(old)
                self._rebuffer(name, view, nbt=True)
        self._flat_w, self._flat_g, self._flat_wt, self._flat_bn, self._flat_nbt = flat_w, flat_g, flat_wt, flat_bn, flat_nbt
        self._dummy = torch.zeros((), device=flat_w.device, requires_grad=True)
        self._ws_pool = []
        self._wt_token = None
        self._attach_grads(zero=True)

    def _rebuffer(self, dotted, tensor, nbt=False):
        parts = dotted.split(".")
        mod = self
        for part in parts[:-1]:
            mod = mod._modules[part]
        mod._buffers[parts[-1]] = tensor
        (self._nbt if nbt else self._bmap)[dotted] = tensor

    def _attach_grads(self, zero):
        if zero:
            self._flat_g.zero_()
        for name, off, shape in self._specs:
            p = self._pmap[name]
            if p.requires_grad:
                p.grad = self._view(self._flat_g, off, shape)

    def _bound_ok(self, device):
        if self._flat_w is None or self._flat_w.device != device:
            return False
        if not self._specs:
            return True
        name, off, shape = self._specs[0]
        return self._pmap[name].data_ptr() == self._flat_w.data_ptr() + off * 4

    def _prepare(self, device):
        if not self._bound_ok(device):
            materialize(self, device)

    def _ensure_grads(self):
        """Before a backward: every trainable parameter's .grad must be its view of the flat gradient buffer.
        zero_grad(set_to_none=True) drops the views of the parameters the optimizer owns (possibly a subset of the
        trainable ones: MultiTaskTrainer's phase 3 trains four modules but optimizes one): those ranges restart from zero,
        the others keep accumulating, as torch's own .grad semantics would have it."""
        first, stale, trainable = True, [], 0
        for name, off, shape in self._specs:
            p = self._pmap[name]
            if not p.requires_grad:
                continue
            trainable += 1
            g = p.grad
            if g is None or (first and g.data_ptr() != self._flat_g.data_ptr() + off * 4):  # (pointer check: first only)
                stale.append((p, off, shape))
            first = False
        if not stale:
            return
        if len(stale) == trainable and trainable == len(self._specs):
            self._attach_grads(zero=True)  # the usual case (optimizer over everything): one memset of the flat buffer
            return
        # a subset: zero the merged ranges (a phase optimizer owning a whole encoder = one memset, not one per tensor; the
        # 64-element alignment padding between tensors holds zeros anyway)
        runs = []
        for p, off, shape in stale:
            n = (math.prod(shape) + 63) // 64 * 64
            if runs and runs[-1][0] + runs[-1][1] == off:
                runs[-1][1] += n
            else:
                runs.append([off, n])
            p.grad = self._view(self._flat_g, off, shape)
        for off, n in runs:
            self._flat_g[off:min(off + n, self._numel)].zero_()

    def _acc_flag(self):
        if self._overwrite_next:
            self._overwrite_next = False
            return 0
        return 1

    def _any_trainable(self):
        return any(p.requires_grad for p in self._pmap.values())

    def _frozen_mask(self):
        """One byte per parameter-table entry (1 = requires_grad False) for the encoder backwards, or None when everything is
        trainable: wholly frozen layers / bottlenecks get no weight-gradient kernels and the backward stops below the lowest
        trainable group (N2: the reference's curriculum phases and fine-tuning freeze sub-graphs)."""
        flags = [0 if self._pmap[n].requires_grad else 1 for n, _, _ in self._specs]
        if not any(flags):
            return None
        return (ctypes.c_uint8 * len(flags))(*flags)

    # ---- gradient-range hook (data parallel) ------------------------------------------------------------------------
    # `_grad_range_hook(eng, offset, length)` (set by FusedTrainStep when world > 1) is called from INSIDE the encoder's
    # backward C call each time the kernels producing grad[offset, offset + length) (engine-relative elements) have been
    # enqueued: a few encoder layers / one ResNet stage at a time, so the all-reduce of that range starts while the rest of
    # the backward is still running. `_grad_ready_hook(eng)` still fires once at the end of the engine's backward.
    def _range_cb(self):
        hook = getattr(self, "_grad_range_hook", None)
        if hook is None:
            return RANGE_CB()  # NULL function pointer: the plain backward
        cb = getattr(self, "_range_cb_obj", None)
        if cb is None or getattr(self, "_range_cb_for", None) is not hook:
            def trampoline(_user, off, length, _self=self):
                h = getattr(_self, "_grad_range_hook", None)
                if h is None:
                    return
                # an engine that was handed a weight-gradient stream announces its ranges as complete on THAT stream
                # (mmsa_resnet_bwd_cb2): make it current, so the reducer records the range's event where the range is produced
                wst = getattr(_self, "_cb_stream", None)
                if wst is not None:
                    with torch.cuda.stream(wst):
                        h(_self, int(off), int(length))
                else:
                    h(_self, int(off), int(length))
            cb = RANGE_CB(trampoline)
            self._range_cb_obj, self._range_cb_for = cb, hook  # keep the ctypes thunk alive
        return cb

    # ---- working copy in the storage dtype ------------------------------------------------------------------------
    def _storage_code(self):
        return MMSA_BF16 if self.precision in ("bf16", "fp8") else MMSA_F32

    def _sync_wt(self):
        """Refresh the bf16 working copy when the fp32 master changed (torch optimizer step, load_state_dict)."""
        if self.precision not in ("bf16", "fp8"):
            return self._flat_w
        token = self._version_token()
        if token != self._wt_token:
            check(_lib.load().mmsa_cast_f32(MMSA_BF16, ptr(self._flat_w), ptr(self._flat_wt), self._numel,
                                            stream_ptr()), "mmsa_cast_f32")
            self._wt_token = token
        return self._flat_wt

    def mark_weights_fresh(self):
        """Called by the fused optimizer, which writes the working copy itself."""
        self._wt_token = self._version_token()

    def _version_token(self):
        """Version counters of EVERY parameter (a torch optimizer over a subset, a partial load_state_dict or an in-place
        edit of one tensor must all refresh the bf16 working copy; ~20 us against a multi-millisecond forward)."""
        return tuple(p._version for p in self._pmap.values())

    # ---- optional side HIP stream ---------------------------------------------------------------------------------
    # The two encoders are independent until the fusion head. With `use_side_stream(True)` this engine enqueues its
    # forward / backward on its own stream (ordered after the caller's stream by an event), so its many small
    # latency-bound kernels (BN finalizes, split-K reduces) overlap with the other encoder's GEMMs. The caller must
    # `join()` before it consumes the result on its own stream (MultiModalEncoder.features and FusedTrainStep do).
    def use_side_stream(self, on):
        self._side = (torch.cuda.Stream(device=self._flat_w.device) if on and self._flat_w is not None
                      and self._flat_w.is_cuda else None)
        self._pending = None

    def use_wgrad_stream(self, on):
        """The image encoder's backward enqueues its stage-wise weight-gradient groups on one more stream (mmsa_resnet_bwd_cb2)."""
        self._wgrad_stream = (torch.cuda.Stream(device=self._flat_w.device) if on and self._flat_w is not None
                              and self._flat_w.is_cuda else None)

    def _run_stream(self):
        """Context manager + stream the engine call goes to."""
        side = getattr(self, "_side", None)
        if side is None:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(side.device))
        side.wait_event(ev)
        return side

    def _mark_pending(self, side):
        ev = torch.cuda.Event()
        ev.record(side)
        self._pending = ev

    def join(self):
        ev = getattr(self, "_pending", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._pending = None

    # ---- per-call workspaces (activations saved for the backward) ---------------------------------------------------
    def _take_ws(self, nbytes, device):
        for i, t in enumerate(self._ws_pool):
            if t.numel() >= nbytes:
                return self._ws_pool.pop(i)
        return torch.empty(int(nbytes), dtype=torch.uint8, device=device)

    def _give_ws(self, ws):
        if len(self._ws_pool) < 2:
            self._ws_pool.append(ws)

    def _apply(self, fn, *a, **k):
        # .to()/.cuda()/.float() replace p.data with fresh tensors: drop the flat binding, it is rebuilt on next use
        out = super()._apply(fn, *a, **k)
        self._flat_w = None
        return out


def engines_of(root):
    return [m for m in root.modules() if isinstance(m, EngineModule)]


def materialize(root, device, precision=None):
    """Pack every engine under `root` into one set of flat buffers on `device`. Returns the FlatState."""
    engs = engines_of(root)
    if precision is not None:
        for e in engs:
            if not isinstance(e, HeadEngine):
                e.precision = precision
    total = sum(e._numel for e in engs)
    bn_total = sum(e._bn_numel for e in engs)
    nbt_total = sum(len(e._nbt) for e in engs)
    flat_w = torch.zeros(total, dtype=torch.float32, device=device)
    flat_g = torch.zeros(total, dtype=torch.float32, device=device)
    need_bf16 = any(e.precision in ("bf16", "fp8") for e in engs)
    flat_wt = torch.zeros(total, dtype=torch.bfloat16, device=device) if need_bf16 else None
    flat_bn = torch.zeros(max(bn_total, 1), dtype=torch.float32, device=device)
    flat_nbt = torch.zeros(max(nbt_total, 1), dtype=torch.long, device=device)
    off = boff = noff = 0
    ranges = []
    for e in engs:
        e.bind(flat_w[off:off + e._numel], flat_g[off:off + e._numel],
               None if flat_wt is None else flat_wt[off:off + e._numel],
               flat_bn[boff:boff + max(e._bn_numel, 0)], flat_nbt[noff:noff + len(e._nbt)])
        ranges.append((e, off, e._numel))
        off += e._numel
        boff += e._bn_numel
        noff += len(e._nbt)
    state = FlatState(root, flat_w, flat_g, flat_wt, flat_bn, flat_nbt, ranges)
    root._flat_state = state
    import weakref
    for e in engs:
        e._owner_state = weakref.ref(state)  # (fused._enclosing_state: an optimizer over a sub-module reuses these buffers)
    return state


class FlatState:
    def __init__(self, root, flat_w, flat_g, flat_wt, flat_bn, flat_nbt, ranges):
        self.root, self.flat_w, self.flat_g, self.flat_wt = root, flat_w, flat_g, flat_wt
        self.flat_bn, self.flat_nbt, self.ranges = flat_bn, flat_nbt, ranges

    def valid(self):
        return all(e._flat_w is not None and e._flat_w.data_ptr() == self.flat_w.data_ptr() + off * 4 and
                   e._bound_ok(self.flat_w.device) for e, off, n in self.ranges)


# ====================================================================================================== encoders
def _bert_cfg(c, batch, seq, out_dim, dtype):
    return BertCfg(batch=batch, seq=seq, hidden=c["hidden"], layers=c["layers"], heads=c["heads"],
                   intermediate=c["intermediate"], vocab=c["vocab"], max_pos=c["max_pos"],
                   type_vocab=c["type_vocab"], out_dim=out_dim, dtype=dtype, ln_eps=c["ln_eps"])


class _BertFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, dummy, ids, mask):
        L = _lib.load()
        B, S = ids.shape
        cfg = _bert_cfg(eng.config, B, S, eng.out_dim, MMSA_FP8 if eng.precision == "fp8" else eng._storage_code())
        nbytes = L.mmsa_bert_ws_bytes(ctypes.byref(cfg))
        if nbytes == 0:
            raise MmsaError(f"unsupported BERT configuration {eng.config} B={B} S={S}")
        ws = eng._take_ws(nbytes, ids.device)
        feat = torch.empty(B, eng.out_dim, dtype=torch.float32, device=ids.device)
        wt = eng._sync_wt()
        check(L.mmsa_bert_fwd(ctypes.byref(cfg), ptr(eng._flat_w), ptr(wt), ptr(ids), ptr(mask), ptr(ws), ptr(feat),
                              stream_ptr()), "mmsa_bert_fwd")
        if dummy is None:
            eng._give_ws(ws)
        else:
            ctx.eng, ctx.cfg, ctx.ws, ctx.ids, ctx.mask, ctx.wt = eng, cfg, ws, ids, mask, wt
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        eng = ctx.eng
        eng._ensure_grads()
        check(_lib.load().mmsa_bert_bwd_cb(ctypes.byref(ctx.cfg), ptr(eng._flat_w), ptr(ctx.wt), ptr(ctx.ids),
                                           ptr(ctx.mask), ptr(ctx.ws), ptr(dfeat.contiguous()), ptr(eng._flat_g),
                                           eng._acc_flag(), stream_ptr(), eng._range_cb(), None,
                                           int(getattr(eng, "layers_per_chunk", 3)), eng._frozen_mask()), "mmsa_bert_bwd")
        eng._give_ws(ctx.ws)
        ctx.ws = None
        if getattr(eng, "_grad_ready_hook", None) is not None:
            eng._grad_ready_hook(eng)
        return None, None, None, None


class BertTextNet(EngineModule):
    """BERT encoder + Linear(hidden, 256): one encoder slot of the reference (MultimodalModel.py:264-266 calls
    `self.*_net(x) -> [B,256]`). NOT in the reference; parameter names follow HF BertModel under `bert.`."""

    def __init__(self, config=None, out_dim=256):
        super().__init__()
        self.config = dict(BERT_BASE if config is None else config)
        self.out_dim = out_dim
        L = _lib.load()
        cfg = _bert_cfg(self.config, 1, 16, out_dim, MMSA_F32)
        specs = param_table(lambda: L.mmsa_bert_param_count(ctypes.byref(cfg)),
                            lambda *a: L.mmsa_bert_param_info(ctypes.byref(cfg), *a))
        self._setup_tables(specs, [], L.mmsa_bert_param_total(ctypes.byref(cfg)), 0)
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():
            for name, p in self._pmap.items():
                if name.startswith("proj."):
                    bound = 1.0 / math.sqrt(self.config["hidden"])
                    p.uniform_(-bound, bound)
                elif "LayerNorm.weight" in name:
                    p.fill_(1.0)
                elif name.endswith("bias"):
                    p.zero_()
                else:
                    p.normal_(0.0, 0.02)  # HF BertPreTrainedModel._init_weights (initializer_range 0.02)

    def forward(self, token_ids, attention_mask=None):
        _require_gpu(token_ids, "BertTextNet")
        self._prepare(token_ids.device)
        ids = token_ids.long().contiguous()  # the reference Trainer applies .float() to its 2nd input (Trainer.py:54)
        mask = None if attention_mask is None else attention_mask.to(torch.float32).contiguous()
        need = torch.is_grad_enabled() and self._any_trainable()
        return _BertFn.apply(self, self._dummy if need else None, ids, mask)


def _resnet_cfg(c, batch, h, w, out_dim, dtype, training):
    cfg = ResnetCfg(batch=batch, height=h, width=w, out_dim=out_dim, dtype=dtype, training=int(training), bn_eps=1e-5,
                    bn_momentum=0.1)
    cfg.blocks[:] = list(c["blocks"])
    cfg.widths[:] = list(c["widths"])
    return cfg


class _ResnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, dummy, image):
        L = _lib.load()
        B, C, H, W = image.shape
        # training: 1 = batch statistics; 0 = eval with everything kept for a backward; 2 = inference (eval, no backward will
        # follow): every BatchNorm folded into its convolution's GEMM epilogue (conv + BN + ReLU (+ residual) as one kernel)
        mode = 1 if eng.training else (2 if dummy is None else 0)
        cfg = _resnet_cfg(eng.config, B, H, W, eng.out_dim, eng._storage_code(), mode)
        nbytes = L.mmsa_resnet_ws_bytes(ctypes.byref(cfg))
        if nbytes == 0 or C != 3:
            raise MmsaError(f"unsupported ResNet input {tuple(image.shape)}")
        ws = eng._take_ws(nbytes, image.device)
        feat = torch.empty(B, eng.out_dim, dtype=torch.float32, device=image.device)
        wt = eng._sync_wt()
        training = eng.training
        side = eng._run_stream()  # (after _sync_wt: a refresh of the working copy is enqueued on the caller's stream)

        def enqueue():
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                check(L.mmsa_resnet_fwd(ctypes.byref(cfg), ptr(eng._flat_w), ptr(wt), ptr(eng._flat_bn), ptr(image), ptr(ws),
                                        ptr(feat), stream_ptr()), "mmsa_resnet_fwd")
                if training:
                    eng._flat_nbt.add_(1)
                if side is not None:
                    eng._mark_pending(side)

        if side is not None:  # allocated on the caller's stream, used on the side stream
            for t in (image, feat, ws):
                t.record_stream(side)
        enqueue()
        if dummy is None:
            eng._give_ws(ws)
        else:
            ctx.eng, ctx.cfg, ctx.ws, ctx.wt = eng, cfg, ws, wt
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        eng = ctx.eng
        eng._ensure_grads()
        dfeat = dfeat.contiguous()
        side = eng._run_stream()
        cfg, ws, wt = ctx.cfg, ctx.ws, ctx.wt
        ctx.ws = None
        acc_flag, cb, frozen = eng._acc_flag(), eng._range_cb(), eng._frozen_mask()
        if side is not None:
            dfeat.record_stream(side)

        def enqueue():
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                # the stage-wise weight-gradient groups on a stream of their own (EngineModule.use_wgrad_stream), beside the
                # latency-bound BatchNorm / data-gradient chain of the following stages; joined before the call returns
                wst = getattr(eng, "_wgrad_stream", None)
                eng._cb_stream = wst  # (see _range_cb: announcements are complete on the weight-gradient stream when there is one)
                check(_lib.load().mmsa_resnet_bwd_cb2(ctypes.byref(cfg), ptr(eng._flat_w), ptr(wt), ptr(ws),
                                                      ptr(dfeat), ptr(eng._flat_g), acc_flag, stream_ptr(),
                                                      ctypes.c_void_p(wst.cuda_stream) if wst is not None else None,
                                                      cb, None, frozen), "mmsa_resnet_bwd")
                eng._give_ws(ws)
                if getattr(eng, "_grad_ready_hook", None) is not None:
                    eng._grad_ready_hook(eng)  # records its event on the stream the gradients are produced on
                if side is not None:
                    eng._mark_pending(side)

        enqueue()
        return None, None, None


class ResNetImageNet(EngineModule):
    """ResNet-50 v1.5 (no fc) + Linear(2048, 256): the other encoder slot. NOT in the reference; parameter names follow
    torchvision under `resnet.` (conv weights logical [O,I,KH,KW], stored channels_last)."""

    def __init__(self, config=None, out_dim=256):
        super().__init__()
        self.config = dict(RESNET50 if config is None else config)
        self.out_dim = out_dim
        L = _lib.load()
        cfg = _resnet_cfg(self.config, 1, 224, 224, out_dim, MMSA_F32, True)
        specs = param_table(lambda: L.mmsa_resnet_param_count(ctypes.byref(cfg), 0),
                            lambda *a: L.mmsa_resnet_param_info(ctypes.byref(cfg), 0, *a))
        bspecs = param_table(lambda: L.mmsa_resnet_param_count(ctypes.byref(cfg), 1),
                             lambda *a: L.mmsa_resnet_param_info(ctypes.byref(cfg), 1, *a))
        self._setup_tables(specs, bspecs, L.mmsa_resnet_param_total(ctypes.byref(cfg), 0),
                           L.mmsa_resnet_param_total(ctypes.byref(cfg), 1))
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():
            for name, p in self._pmap.items():
                if p.dim() == 4:  # torchvision: kaiming_normal_(mode="fan_out", nonlinearity="relu")
                    o, i, kh, kw = p.shape
                    p.normal_(0.0, math.sqrt(2.0 / (o * kh * kw)))
                elif name.startswith("proj."):
                    bound = 1.0 / math.sqrt(self._pmap["proj.weight"].shape[1])
                    p.uniform_(-bound, bound)
                elif name.endswith("weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()

    def forward(self, image):
        _require_gpu(image, "ResNetImageNet")
        self._prepare(image.device)
        img = image.to(torch.float32).contiguous()
        need = torch.is_grad_enabled() and self._any_trainable()
        return _ResnetFn.apply(self, self._dummy if need else None, img)


# ====================================================================================================== head modules
class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, dummy, *inputs):
        L = _lib.load()
        B = inputs[0].shape[0]
        cfg = eng._head_cfg(B)
        nbytes = L.mmsa_head_ws_bytes(eng.kind, ctypes.byref(cfg))
        if nbytes == 0:
            raise MmsaError(f"unsupported head configuration kind={eng.kind} B={B}")
        dev = inputs[0].device
        ws = eng._take_ws(nbytes, dev)
        outs = [torch.empty(B, n, dtype=torch.float32, device=dev) for n in eng._out_dims()]
        ins = ptr_array(inputs)
        check(L.mmsa_head_fwd(eng.kind, ctypes.byref(cfg), ptr(eng._flat_w), ptr(eng._flat_bn), ins, ptr_array(outs), ptr(ws),
                              stream_ptr()), "mmsa_head_fwd")
        if eng.training and len(eng._nbt):
            eng._flat_nbt.add_(1)
        if dummy is not None or any(ctx.needs_input_grad[2:]):
            ctx.eng, ctx.cfg, ctx.ws, ctx.inputs, ctx.outs = eng, cfg, ws, inputs, outs
        else:
            eng._give_ws(ws)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        eng = ctx.eng
        L = _lib.load()
        train_params = eng._any_trainable()
        if train_params:
            eng._ensure_grads()
        dins = [torch.empty_like(t) for t in ctx.inputs]
        d = [None if g is None else g.contiguous() for g in douts]
        if d[0] is None:
            d[0] = torch.zeros_like(ctx.outs[0])
        if eng.kind == HEAD_CLASSIFIER and d[1] is None:
            d[1] = torch.zeros_like(ctx.outs[1])
        check(L.mmsa_head_bwd(eng.kind, ctypes.byref(ctx.cfg), ptr(eng._flat_w), ptr_array(ctx.inputs), ptr_array(d),
                              ptr_array(dins), ptr(eng._flat_g) if train_params else None,
                              eng._acc_flag() if train_params else 0, ptr(ctx.ws), stream_ptr()), "mmsa_head_bwd")
        eng._give_ws(ctx.ws)
        ctx.ws = None
        if getattr(eng, "_grad_ready_hook", None) is not None:
            eng._grad_ready_hook(eng)
        return (None, None) + tuple(dins)


class HeadEngine(EngineModule):
    """fp32 fusion-head module backed by mmsa_head_fwd / mmsa_head_bwd."""

    kind = -1
    precision = "fp32"
    _seed_counter = 0

    def _base_cfg(self):
        return dict(embed=256, tokens=1, heads=1, pool_mode=0, num_classes=3, valence=0, hidden=128, out_dim=128,
                    dropout_p=0.0)

    def _init_head(self):
        L = _lib.load()
        cfg = self._head_cfg(1)
        k = self.kind
        specs = param_table(lambda: L.mmsa_head_param_count(k, ctypes.byref(cfg), 0),
                            lambda *a: L.mmsa_head_param_info(k, ctypes.byref(cfg), 0, *a))
        bspecs = param_table(lambda: L.mmsa_head_param_count(k, ctypes.byref(cfg), 1),
                             lambda *a: L.mmsa_head_param_info(k, ctypes.byref(cfg), 1, *a))
        self._setup_tables(specs, bspecs, L.mmsa_head_param_total(k, ctypes.byref(cfg), 0),
                           L.mmsa_head_param_total(k, ctypes.byref(cfg), 1))
        self.reset_parameters()

    def _head_cfg(self, batch):
        c = self._base_cfg()
        HeadEngine._seed_counter += 1
        seed = (torch.initial_seed() * 1000003 + HeadEngine._seed_counter) & 0xFFFFFFFFFFFFFFFF
        return HeadCfg(batch=batch, embed=c["embed"], tokens=c["tokens"], heads=c["heads"], pool_mode=c["pool_mode"],
                       num_classes=c["num_classes"], valence=c["valence"], hidden=c["hidden"], out_dim=c["out_dim"],
                       training=int(self.training), bn_eps=1e-5, bn_momentum=0.1, ln_eps=1e-5,
                       dropout_p=float(c["dropout_p"]), seed=seed)

    def reset_parameters(self):
        """torch defaults: nn.Linear kaiming_uniform(a=sqrt 5) == U(+-1/sqrt(fan_in)) for weight and bias;
        nn.MultiheadAttention: xavier_uniform in_proj, zero in_proj/out_proj bias; norms weight 1 / bias 0."""
        with torch.no_grad():
            for name, p in self._pmap.items():
                if name.endswith("in_proj_weight"):
                    nn.init.xavier_uniform_(p)
                elif name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
                    p.zero_()
                elif p.dim() == 2:
                    bound = 1.0 / math.sqrt(p.shape[1])
                    p.uniform_(-bound, bound)
                    bname = name[:-len("weight")] + "bias"
                    if bname in self._pmap:
                        self._pmap[bname].uniform_(-bound, bound)
                elif p.dim() == 1 and name.endswith("weight"):
                    p.fill_(1.0)  # BatchNorm / LayerNorm scale
                elif p.dim() == 1 and (name[:-len("bias")] + "weight") in self._pmap and \
                        self._pmap[name[:-len("bias")] + "weight"].dim() == 1:
                    p.zero_()  # BatchNorm / LayerNorm shift

    def _run(self, *inputs):
        for t in inputs:
            _require_gpu(t, type(self).__name__)
        self._prepare(inputs[0].device)
        ins = [t.to(torch.float32).contiguous() for t in inputs]
        need = torch.is_grad_enabled() and self._any_trainable()
        return _HeadFn.apply(self, self._dummy if need else None, *ins)


class CrossEntropyFn(torch.autograd.Function):
    """Fused CE forward + dlogits (mmsa_ce_fwd_bwd): nn.CrossEntropyLoss() of Trainer.py:17,68."""

    @staticmethod
    def forward(ctx, logits, labels):
        _require_gpu(logits, "CrossEntropy")
        lg = logits.to(torch.float32).contiguous()
        lb = labels.long().contiguous()
        B, C = lg.shape
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        dlogits = torch.empty_like(lg)
        check(_lib.load().mmsa_ce_fwd_bwd(ptr(lg), ptr(lb), ptr(loss), ptr(dlogits), None, B, C, 1.0, stream_ptr()),
              "mmsa_ce_fwd_bwd")
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None


class CrossEntropyLoss(nn.Module):
    """Drop-in for `nn.CrossEntropyLoss()` (mean reduction, int64 targets) on the fused kernel."""

    def forward(self, logits, labels):
        return CrossEntropyFn.apply(logits, labels)


# ------------------------------------------------------------------------------------------------ N1: contrastive losses
def _contrastive_ws(B, D, device):
    from .kernels import workspace
    return workspace(_lib.load().mmsa_contrastive_ws_bytes(B, D), device, "contrastive")


class InfoNCEFn(torch.autograd.Function):
    """Fused forward + backward of the reference's supervised InfoNCE with learnable temperature
    (`MultimodalTransformerModel.compute_contrastive_loss`, MultimodalModel.py:232-260): mmsa_infonce_fwd_bwd."""

    @staticmethod
    def forward(ctx, feat1, feat2, labels, temperature):
        _require_gpu(feat1, "InfoNCE")
        same = feat1 is feat2 or (feat1.data_ptr() == feat2.data_ptr() and feat1.shape == feat2.shape)
        f1 = feat1.to(torch.float32).contiguous()
        f2 = f1 if same else feat2.to(torch.float32).contiguous()
        lb = labels.long().contiguous()
        t = temperature.detach().to(torch.float32).reshape(-1)[:1].contiguous()
        B, D = f1.shape
        loss = torch.empty((), dtype=torch.float32, device=f1.device)
        d1, d2 = torch.empty_like(f1), torch.empty_like(f1)
        dt = torch.empty(1, dtype=torch.float32, device=f1.device)
        ws = _contrastive_ws(B, D, f1.device)
        check(_lib.load().mmsa_infonce_fwd_bwd(ptr(f1), ptr(f2), ptr(lb), ptr(t), ptr(loss), ptr(d1), ptr(d2), ptr(dt), B, D,
                                               1.0, ptr(ws), stream_ptr()), "mmsa_infonce_fwd_bwd")
        ctx.same = same
        ctx.tshape = temperature.shape
        ctx.save_for_backward(d1, d2, dt)
        return loss

    @staticmethod
    def backward(ctx, g):
        d1, d2, dt = ctx.saved_tensors
        # both arguments were the same tensor: autograd adds the two returned gradients
        return d1 * g, d2 * g, None, (dt * g).reshape(ctx.tshape)


class SupConFn(torch.autograd.Function):
    """Fused forward + backward of the two-view supervised contrastive loss (`contrastive_loss`, train.py:16-40)."""

    @staticmethod
    def forward(ctx, z1, z2, labels, temperature):
        _require_gpu(z1, "SupCon")
        a = z1.to(torch.float32).contiguous()
        b = z2.to(torch.float32).contiguous()
        lb = labels.long().reshape(-1).contiguous()
        B, D = a.shape
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        d1, d2 = torch.empty_like(a), torch.empty_like(b)
        ws = _contrastive_ws(B, D, a.device)
        check(_lib.load().mmsa_supcon_fwd_bwd(ptr(a), ptr(b), ptr(lb), float(temperature), ptr(loss), ptr(d1), ptr(d2), B, D,
                                              1.0, ptr(ws), stream_ptr()), "mmsa_supcon_fwd_bwd")
        ctx.save_for_backward(d1, d2)
        return loss

    @staticmethod
    def backward(ctx, g):
        d1, d2 = ctx.saved_tensors
        return d1 * g, d2 * g, None, None


class _GatherRows(torch.autograd.Function):
    """Data-parallel contrastive terms: the rows of every rank, concatenated in rank order, so that the
    negatives of an anchor are the whole GLOBAL batch, as in the reference's single process (MultimodalModel.py:232-260,
    train.py:16-40). Forward: all-gather (RCCL over xGMI; gloo in the CPU tests). Every rank then computes the same global loss,
    so the gradient of that loss w.r.t. this rank's rows is its slice of the full gradient — no collective in the backward —
    times world_size, because the trainer averages parameter gradients over ranks (sum / world) and each rank contributes only
    the part that flows through its own rows."""

    @staticmethod
    def forward(ctx, x, group):
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        x = x.contiguous()
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x, group=group)
        ctx.world, ctx.rank, ctx.rows = world, rank, x.shape[0]
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        a = ctx.rank * ctx.rows
        return g[a:a + ctx.rows] * float(ctx.world), None


def global_rows(x, group=None):
    """x [B, ...] on every rank -> [world * B, ...] in rank order (identity without an initialised process group). Differentiable
    for floating-point inputs; labels (integers) are gathered without a graph."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x
    if x.is_floating_point() and x.requires_grad:
        return _GatherRows.apply(x, group)
    x = x.contiguous()
    parts = [torch.empty_like(x) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, x, group=group)
    return torch.cat(parts, 0)


def supervised_infonce(feat1, feat2, labels, temperature, global_negatives=True):
    """loss = InfoNCE(feat1, feat2 | labels, temperature) — MultimodalModel.py:232-260 on the fused kernel. Under data parallelism
    the features and labels of all ranks are gathered first (global_negatives), so the loss is that of the global batch."""
    if global_negatives:
        same = feat1 is feat2
        feat1 = global_rows(feat1)
        feat2 = feat1 if same else global_rows(feat2)
        labels = global_rows(labels)
    return InfoNCEFn.apply(feat1, feat2, labels, temperature)


def supcon_loss(z1, z2, labels, temperature=0.1, global_negatives=True):
    """train.py:16-40 on the fused kernel (two views; global batch under data parallelism)."""
    if global_negatives:
        z1, z2, labels = global_rows(z1), global_rows(z2), global_rows(labels)
    return SupConFn.apply(z1, z2, labels, temperature)


def nt_xent_loss(z1, z2, temperature=0.5, global_negatives=True):
    """The label-free NT-Xent of the reference's ME-MHACL script (ME-MHACL/train.py:47-66): cross-entropy of the [2B, 2B] cosine
    similarity / T with the diagonal masked out, target = the other view of the same sample. That is the two-view supervised
    contrastive loss (train.py:16-40) with every sample its own class — one positive per anchor — so it runs on the same fused
    kernel with labels = arange(B); the two differ only by train.py's +1e-8 guards (below fp32 resolution of the terms here)."""
    if global_negatives:
        z1, z2 = global_rows(z1), global_rows(z2)
    labels = torch.arange(z1.shape[0], device=z1.device)
    return SupConFn.apply(z1, z2, labels, temperature)
