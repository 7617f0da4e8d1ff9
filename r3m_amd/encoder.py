"""ResNet-18/34/50 frame encoder on the HIP engine (libr3m_hip.so), behind an nn.Module that has the state-dict of
torchvision.models.resnet{18,34,50} with fc = Identity — what the reference builds at
/root/reference/r3m/models/models_r3m.py:44-52,62-63 and calls at :99.

Design (MI355X-first, not a translation of torchvision's module graph):
  * the network is ONE native plan in C++ (r3m_amd/csrc/engine.hip): forward and backward are single C calls that enqueue
    the whole kernel sequence on the current HIP stream; Python owns only memory (torch allocator) and autograd glue;
  * parameters are views into ONE flat fp32 buffer (torchvision order), gradients views into a second one: the fused Adam
    step and the RCCL gradient all-reduce work on contiguous slices, no per-tensor loops, no bucket copies;
  * conv weights are logical OIHW tensors with channels_last strides == the OHWI image the kernels read, so
    load_state_dict()/state_dict() interchange with reference checkpoints needs no layout conversion.

There is no CPU or eager fallback: forward() on a non-CUDA tensor raises.
"""
import ctypes as C
import math

import weakref
import torch
import torch.nn as nn

from . import _lib


class _Node(nn.Module):
    """Anonymous container: exists only so parameter names nest like torchvision's (layer1.0.downsample.0.weight)."""


def _tensor_table(size):
    """[(name, kind, offset, shape)] from the native plan (torchvision state-dict order)."""
    L = _lib.lib()
    h = L.r3m_resnet_create(size, 1)
    if not h:
        raise ValueError(_lib.last_error())
    try:
        out = []
        name = C.create_string_buffer(128)
        kind, ndim = C.c_int(), C.c_int()
        off = C.c_longlong()
        shape = (C.c_int * 4)()
        for i in range(L.r3m_resnet_num_tensors(h)):
            _lib.check(L.r3m_resnet_tensor_info(h, i, name, 128, C.byref(kind), C.byref(off), C.byref(ndim), shape), "tensor_info")
            out.append((name.value.decode(), kind.value, off.value, tuple(shape[k] for k in range(ndim.value))))
        return out, L.r3m_resnet_num_params(h), L.r3m_resnet_num_buffers(h), L.r3m_resnet_out_dim(h)
    finally:
        L.r3m_resnet_destroy(h)


class _EncoderFn(torch.autograd.Function):
    """h = encoder(x). Backward writes parameter gradients straight into the module's flat gradient buffer."""

    @staticmethod
    def forward(ctx, x, anchor, module, training, crop=None):
        h = module._run_forward(x, training, crop, saved=True)
        ctx.module = module
        ctx.slot, ctx.generation = module._last_forward
        if 0 <= ctx.slot < len(module._ring):
            # the slot stays "live" while this graph node exists and has not run its backward: forwards that need no backward
            # (no_grad / eval calls between a training forward and its backward) must not take it (_pick_slot)
            module._ring[ctx.slot].waiting = weakref.ref(ctx)
        return h

    @staticmethod
    def backward(ctx, dh):
        m = ctx.module
        # The data-parallel stage hook starts an ASYNC all-reduce on a slice of the flat gradient buffer. With several live forwards
        # ((h1 + h2).backward()) every backward accumulates into the same buffer, so only the LAST outstanding one may fire it: an
        # earlier one would hand RCCL a slice the next backward is still adding to. "Outstanding" = ring slots whose autograd node is
        # still alive and has not run (weak references: a forward whose graph was dropped — an eval call without no_grad, an
        # exception before backward — stops counting when its node is collected; round 5 kept a manual counter that such a forward
        # left raised for good, silently costing every later step its comm / compute overlap: ADVICE r5).
        others = sum(1 for sl in m._ring if sl.live and sl.waiting() is not ctx)
        m._run_backward(dh.contiguous(), ctx.generation, ctx.slot, fire_hooks=others == 0)
        if 0 <= ctx.slot < len(m._ring):
            w = m._ring[ctx.slot].waiting
            if w is not None and w() is ctx:
                m._ring[ctx.slot].waiting = None
        return None, None, None, None, None


PRECISIONS = {"fp32": 0, "bf16": 1}   # R3M_DT_F32 / R3M_DT_BF16 (include/r3m_hip.h)


class _LiveSlot:
    """Saved state of one forward pass: native plans (per frame count; a plan also carries the staged-backward state), the HBM
    arena holding that forward's activations, and a generation counter that tells a late backward its activations are gone."""
    __slots__ = ("plans", "arena", "generation", "F", "waiting")

    def __init__(self):
        self.plans, self.arena, self.generation, self.F = {}, None, 0, None
        self.waiting = None      # weakref to the autograd node whose backward will read this slot's activations (None: nobody)

    @property
    def live(self):
        return self.waiting is not None and self.waiting() is not None


def _pick_slot(live, pos, saved):
    """Which slot the next forward writes its activations to. live[i]: slot i holds a forward whose backward is still to come;
    pos: round-robin cursor; saved: this forward goes through autograd (a backward will read it). Returns (slot, new cursor);
    slot -1 = the scratch slot (forwards nobody differentiates, when every ring slot is live).
      * a free slot is always preferred (cursor order), so an inference call between a training forward and its backward never
        evicts the training forward;
      * a SAVED forward with every slot live evicts the oldest (the cursor): the evicted forward's backward raises and names the
        remedy (max_live_forwards) — unchanged behaviour;
      * an UNSAVED forward with every slot live runs in the scratch slot (its arena is allocated on first use and kept;
        HipResNet.release_scratch() hands the HBM back)."""
    k = len(live)
    order = [(pos + i) % k for i in range(k)]
    for i in order:
        if not live[i]:
            return i, ((i + 1) % k if saved else pos)
    if saved:
        return order[0], (order[0] + 1) % k
    return -1, pos


class HipResNet(nn.Module):
    """precision="fp32": everything fp32 (the reference's arithmetic). precision="bf16": activations and their gradients are
    stored in bf16 and the convolutions run on the bf16 MFMA with fp32 accumulation; parameters, their gradients, BatchNorm
    statistics and the output embedding stay fp32 (what torch.autocast(bfloat16) around the reference's encoder would do)."""

    def __init__(self, size, precision="fp32", max_live_forwards=1):
        """max_live_forwards: how many forward passes may be waiting for their backward at once. The reference module is a plain
        autograd graph (/root/reference/r3m/models/models_r3m.py:84-100): h1 = enc(x1); h2 = enc(x2); (h1 + h2).sum().backward()
        works there. Here a forward's saved activations live in ONE preallocated HBM arena per slot (141 GB for ResNet-50 fp32 at
        1280 frames), so the default of 1 — all `Trainer.update` needs — makes a second forward invalidate the first one's backward
        (it raises, it never computes on overwritten activations); k > 1 keeps a ring of k arenas + plans, forwards take them
        round-robin, and any k consecutive forwards can be backpropagated in any order."""
        super().__init__()
        if precision not in PRECISIONS:
            raise ValueError(f"HipResNet: precision {precision!r} (expected one of {sorted(PRECISIONS)})")
        if int(max_live_forwards) < 1:
            raise ValueError("HipResNet: max_live_forwards must be >= 1")
        self.max_live_forwards = int(max_live_forwards)
        self.precision = precision
        table, n_params, n_buffers, out_dim = _tensor_table(size)
        self.size = size
        self.outdim = out_dim
        self._n_params = n_params
        self._n_buffers = n_buffers
        self._table = table
        self._slots = []        # (tensor, kind, offset, shape) in table order, for re-flattening
        flat_p = torch.zeros(n_params, dtype=torch.float32)
        flat_b = torch.zeros(n_buffers, dtype=torch.float32)
        n_bn = sum(1 for t in table if t[1] == 1)
        flat_nbt = torch.zeros(n_bn, dtype=torch.int64)
        bn_i = 0
        for name, kind, off, shape in table:
            parent, leaf = self._resolve(name)
            n = int(math.prod(shape))
            if kind == 0:
                O, I, kh, kw = shape
                t = nn.Parameter(flat_p[off:off + n].view(O, kh, kw, I).permute(0, 3, 1, 2))  # logical OIHW, physical OHWI
                parent.register_parameter(leaf, t)
            elif kind in (1, 2):
                t = nn.Parameter(flat_p[off:off + n].view(shape))
                parent.register_parameter(leaf, t)
            else:
                t = flat_b[off:off + n].view(shape)
                parent.register_buffer(leaf, t)
                if kind == 4:
                    parent.register_buffer("num_batches_tracked", flat_nbt[bn_i])
                    bn_i += 1
        self.fc = nn.Identity()   # models_r3m.py:62
        self._flat_p, self._flat_b, self._flat_nbt = flat_p, flat_b, flat_nbt
        self._flat_g = None
        self._ring = [_LiveSlot() for _ in range(self.max_live_forwards)]   # _plans / _arena below are slot 0's
        self._ring_pos = 0
        self._scratch = None          # _LiveSlot for forwards without a backward while every ring slot is live (_pick_slot)
        self._last_forward = (0, 0)   # (slot, generation) of the most recent forward
        self._grad_fresh = True
        self._stage_hook = None   # callable(stage, offset, count) after each backward stage (data-parallel wrapper)
        self.reset_parameters()

    @property
    def _plans(self):             # F -> native handle (slot 0)
        return self._ring[0].plans

    @_plans.setter
    def _plans(self, v):
        self._ring[0].plans = v

    @property
    def _arena(self):
        return self._ring[0].arena

    @_arena.setter
    def _arena(self, v):          # assigning None drops EVERY slot's arena (re-flatten / device move)
        if v is None:
            for sl in self._ring:
                sl.arena = None
            if self._scratch is not None:
                self._scratch.arena = None
        else:
            self._ring[0].arena = v

    # ---- structure -------------------------------------------------------------------------------------------
    def _resolve(self, dotted):
        parts = dotted.split(".")
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Node())
            mod = mod._modules[p]
        return mod, parts[-1]

    def _named_slots(self):
        """(name, kind, offset, shape, tensor) following the native table."""
        sd_p = dict(self.named_parameters())
        sd_b = dict(self.named_buffers())
        for name, kind, off, shape in self._table:
            yield name, kind, off, shape, (sd_p[name] if kind <= 2 else sd_b[name])

    def reset_parameters(self):
        """torchvision ResNet init: kaiming_normal_(fan_out, relu) on convs, BN weight 1 / bias 0.
        Draws from the global torch RNG in state-dict order, like torchvision's `for m in self.modules()` loop."""
        with torch.no_grad():
            for name, kind, off, shape, t in self._named_slots():
                if kind == 0:
                    w = torch.empty(shape, dtype=torch.float32)
                    nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
                    t.copy_(w)
                elif kind == 1 or kind == 4:
                    t.fill_(1.0)
                else:
                    t.zero_()
            self._flat_nbt.zero_()

    @property
    def _awaiting(self):
        """forwards through autograd whose backward has not run yet (see _EncoderFn.backward)"""
        return sum(1 for sl in self._ring if sl.live)

    # ---- flat storage ----------------------------------------------------------------------------------------
    def _is_flat(self):
        base = self._flat_p.data_ptr()
        dev = self._flat_p.device
        for name, kind, off, shape, t in self._named_slots():
            if kind <= 2 and (t.device != dev or t.data_ptr() != base + off * 4):
                return False
            if kind > 2 and (t.device != self._flat_b.device or t.data_ptr() != self._flat_b.data_ptr() + off * 4):
                return False
        return True

    def _reflatten(self):
        """Re-establish 'every tensor is a view of the flat buffers' after .to()/.cuda()/deepcopy/load with assign."""
        slots = list(self._named_slots())
        dev = slots[0][4].device
        for _, _, _, _, t in slots:
            if t.dtype != torch.float32:
                raise TypeError("r3m_amd.HipResNet keeps fp32 master parameters; dtype conversion is not supported")
        flat_p = torch.zeros(self._n_params, dtype=torch.float32, device=dev)
        flat_b = torch.zeros(self._n_buffers, dtype=torch.float32, device=dev)
        flat_nbt = torch.zeros_like(self._flat_nbt, device=dev)
        bn_i = 0
        with torch.no_grad():
            for name, kind, off, shape, t in slots:
                n = int(math.prod(shape))
                if kind == 0:
                    O, I, kh, kw = shape
                    v = flat_p[off:off + n].view(O, kh, kw, I).permute(0, 3, 1, 2)
                elif kind in (1, 2):
                    v = flat_p[off:off + n].view(shape)
                else:
                    v = flat_b[off:off + n].view(shape)
                v.copy_(t)
                if kind <= 2:
                    t.data = v
                    t.grad = None
                else:
                    parent, leaf = self._resolve(name)
                    parent._buffers[leaf] = v
                    if kind == 4:
                        old = parent._buffers["num_batches_tracked"]
                        flat_nbt[bn_i] = old.to(dev)
                        parent._buffers["num_batches_tracked"] = flat_nbt[bn_i]
                        bn_i += 1
        self._flat_p, self._flat_b, self._flat_nbt = flat_p, flat_b, flat_nbt
        self._flat_g = None
        self._arena = None
        self._grad_fresh = True

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self._reflatten()
        return self

    def flat_params(self):
        self._ensure()
        return self._flat_p

    def flat_grads(self):
        self._ensure()
        if self._flat_g is None:
            self._flat_g = torch.zeros_like(self._flat_p)
            for name, kind, off, shape, t in self._named_slots():
                if kind == 0:
                    O, I, kh, kw = shape
                    t.grad = self._flat_g[off:off + t.numel()].view(O, kh, kw, I).permute(0, 3, 1, 2)
                elif kind <= 2:
                    t.grad = self._flat_g[off:off + t.numel()].view(shape)
        return self._flat_g

    def _ensure(self):
        if not self._is_flat():
            self._reflatten()

    def mark_grads_stale(self):
        """Called by the optimizer's zero_grad(): the next backward overwrites instead of accumulating."""
        self._grad_fresh = True

    def stage_range(self, stage):
        L = _lib.lib()
        h = self._plan(1)
        off, cnt = C.c_longlong(), C.c_longlong()
        _lib.check(L.r3m_resnet_stage_range(h, stage, C.byref(off), C.byref(cnt)), "stage_range")
        return off.value, cnt.value

    # ---- execution -------------------------------------------------------------------------------------------
    def _slot(self, si):
        if si >= 0:
            return self._ring[si]
        if self._scratch is None:
            self._scratch = _LiveSlot()
        return self._scratch

    def release_scratch(self):
        """Drop the scratch arena (HBM held by no_grad forwards that ran while every saved forward was still awaiting its backward)."""
        if self._scratch is not None:
            self._scratch.arena = None

    def _plan(self, F, slot=0):
        plans = self._slot(slot).plans
        h = plans.get(F)
        if h is None:
            h = _lib.lib().r3m_resnet_create_dt(self.size, F, PRECISIONS[self.precision])
            if not h:
                raise RuntimeError(_lib.last_error())
            plans[F] = h
        return h

    def __del__(self):
        try:
            L = _lib.lib()
            for sl in self._ring + ([self._scratch] if self._scratch is not None else []):
                for h in sl.plans.values():
                    L.r3m_resnet_destroy(h)
        except Exception:
            pass

    def __getstate__(self):
        """copy.deepcopy / pickle: native plan handles, the activation arena, the flat gradient buffer and the data-parallel
        hook belong to THIS object (a copied integer handle would be destroyed twice); the copy re-creates them lazily. The
        reference R3M deep-copies cleanly (plain nn.Module), so must this."""
        st = self.__dict__.copy()
        st.update(_ring=[_LiveSlot() for _ in self._ring], _ring_pos=0, _scratch=None, _last_forward=(0, 0), _flat_g=None,
                  _stage_hook=None, _grad_fresh=True)
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        for p in self.parameters():      # gradients were views of the dropped flat buffer
            p.grad = None

    def _run_forward(self, x, training, crop=None, saved=False):
        """x: [F,3,224,224] fp32 frames, or None with crop = augment.CroppedClips (raw clips + boxes, resampled in the stem pre-pass).
        saved: the call comes from autograd (_EncoderFn) and a backward will read this forward's activations."""
        L = _lib.lib()
        src = crop.raw if crop is not None else x
        F = src.shape[0]
        si, self._ring_pos = _pick_slot([sl.live for sl in self._ring], self._ring_pos, saved)
        slot = self._slot(si)
        slot.waiting = None                              # whatever forward was saved here is gone now
        h = self._plan(F, si)
        need = L.r3m_resnet_arena_bytes(h)
        if slot.arena is None or slot.arena.numel() < need or slot.arena.device != src.device:
            slot.arena = None   # release first: the arena is the dominant HBM allocation
            slot.arena = torch.empty(need, dtype=torch.uint8, device=src.device)
        arena = slot.arena
        out = torch.empty((F, self.outdim), dtype=torch.float32, device=src.device)
        # 1 train / 0 eval with the activations kept for a backward / 2 inference: eval statistics and nothing kept — BatchNorm, residual
        # join and ReLU ride in the convolutions' stores (what load_r3m(...).eval() under no_grad runs, /root/reference/r3m/__init__.py:72-75)
        mode = 1 if training else (0 if saved else 2)
        with _lib.on(src):
            if crop is not None:
                _lib.check(L.r3m_resnet_forward_crop(h, crop.raw.data_ptr(), 1 if crop.raw.dtype == torch.uint8 else 0,
                                                     crop.boxes.data_ptr(), crop.frames_per_box, crop.raw.shape[-2], crop.raw.shape[-1],
                                                     self._flat_p.data_ptr(), self._flat_b.data_ptr(), arena.data_ptr(),
                                                     out.data_ptr(), mode, _lib.stream_ptr(src.device)),
                           "resnet_forward_crop")
            else:
                _lib.check(L.r3m_resnet_forward(h, x.data_ptr(), self._flat_p.data_ptr(), self._flat_b.data_ptr(),
                                                arena.data_ptr(), out.data_ptr(), mode,
                                                _lib.stream_ptr(x.device)), "resnet_forward")
        if training:
            self._flat_nbt += 1
        slot.generation += 1
        slot.F = F
        self._last_forward = (si, slot.generation)
        return out

    def _run_backward(self, dh, generation, si=0, fire_hooks=True):
        slot = self._slot(si)
        if generation != slot.generation or slot.arena is None:
            raise RuntimeError(f"r3m_amd: the encoder ran {len(self._ring)} other forward(s) before this backward; its saved "
                               f"activations (one HBM arena per live forward) were overwritten. Construct the encoder with "
                               f"max_live_forwards=k (R3M(..., max_live_forwards=k)) to keep k forwards alive at once")
        L = _lib.lib()
        h = self._plan(slot.F, si)
        g = self.flat_grads()
        accumulate = 0 if self._grad_fresh else 1
        with _lib.on(dh):
            for stage in range(4):
                _lib.check(L.r3m_resnet_backward(h, dh.data_ptr(), self._flat_p.data_ptr(), g.data_ptr(), slot.arena.data_ptr(), stage,
                                                 stage + 1, accumulate, _lib.stream_ptr(dh.device)), "resnet_backward")
                if self._stage_hook is not None and fire_hooks:
                    off, cnt = self.stage_range(stage)
                    self._stage_hook(stage, off, cnt)
        self._grad_fresh = False

    def forward(self, x):
        """x: [F,3,224,224] float32 CUDA tensor with values in 0..255 (the /255 and Normalize of R3M.forward are fused into
        the stem kernel). Returns [F, outdim]."""
        if not x.is_cuda:
            raise RuntimeError("r3m_amd: the encoder runs on MI355X through libr3m_hip.so only; got a CPU tensor "
                               "(no CPU / eager fallback exists — the CPU oracle lives under oracle/ for tests)")
        from .augment import CroppedClips
        crop = x if isinstance(x, CroppedClips) else None
        if crop is not None and tuple(crop.out_hw) != (224, 224):
            raise ValueError(f"CroppedClips must resample to 224x224, got {crop.out_hw}")
        if crop is None and (x.dim() != 4 or tuple(x.shape[1:]) != (3, 224, 224)):
            raise ValueError(f"expected [F,3,224,224], got {tuple(x.shape)}")
        self._ensure()
        if self._flat_p.device != x.device:
            raise RuntimeError(f"encoder parameters on {self._flat_p.device}, input on {x.device}")
        if crop is None:
            x = x.contiguous()
            if x.dtype != torch.float32:
                x = x.float()
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if needs_grad:
            anchor = next(self.parameters())
            return _EncoderFn.apply(None if crop is not None else x, anchor, self, self.training, crop)
        return self._run_forward(None if crop is not None else x, self.training, crop)
