"""Fused Adam over flat parameter buffers (one HIP kernel per flat buffer) — the `encoder_opt` of R3M.

Replaces torch.optim.Adam(params, lr=lr) built at /root/reference/r3m/models/models_r3m.py:76 and stepped at
/root/reference/r3m/trainer.py:156-158 (defaults: betas (0.9, 0.999), eps 1e-8, weight_decay 0, amsgrad False).
"""
import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    """`owners` are modules exposing flat_params() / flat_grads() / mark_grads_stale() (HipResNet, LanguageReward)."""

    def __init__(self, owners, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.owners = list(owners)
        params = [p for o in self.owners for p in o.parameters()]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._steps = [0] * len(self.owners)   # torch.optim.Adam keeps `step` per parameter: an owner that receives its first
        self._m = [None] * len(self.owners)    # gradient later (the language head) starts its bias correction then
        self._v = [None] * len(self.owners)
        self.grad_scale = 1.0

    def __getstate__(self):
        # Optimizer.__getstate__ keeps defaults/state/param_groups only; deepcopy(R3M) and pickling must keep the owners
        # protocol and the moments too (the reference's torch.optim.Adam deep-copies with its state).
        st = super().__getstate__()
        st.update(owners=self.owners, _steps=self._steps, _m=self._m, _v=self._v, grad_scale=self.grad_scale)
        return st

    def zero_grad(self, set_to_none=True):
        # Gradients live in persistent flat buffers; "zeroing" = the next backward overwrites them (no memset pass).
        for o in self.owners:
            o.mark_grads_stale()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FusedAdam.step(closure) is not supported")
        g0 = self.param_groups[0]
        lr, (b1, b2), eps = g0["lr"], g0["betas"], g0["eps"]
        L = _lib.lib()
        for i, o in enumerate(self.owners):
            p = o.flat_params()
            if not p.is_cuda:
                raise RuntimeError("r3m_amd.FusedAdam: parameters must live on the GPU (HIP kernel, no CPU fallback)")
            if not getattr(o, "has_grads", lambda: True)():
                continue
            g = o.flat_grads()
            if self._m[i] is None or self._m[i].device != p.device or self._m[i].numel() != p.numel():
                self._m[i] = torch.zeros_like(p)
                self._v[i] = torch.zeros_like(p)
            n = p.numel()
            pad = (-n) % 4
            assert pad == 0, "flat buffers are padded to multiples of 4 floats"
            self._steps[i] += 1
            with _lib.on(p):
                _lib.check(L.r3m_adam_step(p.data_ptr(), g.data_ptr(), self._m[i].data_ptr(), self._v[i].data_ptr(), n, float(lr),
                                           float(b1), float(b2), float(eps), self._steps[i], float(self.grad_scale),
                                           _lib.stream_ptr(p.device)), "adam_step")

    def moments(self, param):
        """Read-only views (exp_avg, exp_avg_sq) of Adam's moments for one parameter, shaped and strided like it — what
        `torch.optim.Adam.state[param]` holds in the reference (models_r3m.py:76). None before the owner's first step."""
        for i, o in enumerate(self.owners):
            flat = o.flat_params()
            off = (param.data_ptr() - flat.data_ptr()) // flat.element_size()
            if param.device == flat.device and 0 <= off and off + param.numel() <= flat.numel():
                if self._m[i] is None:
                    return None
                view = lambda t: torch.as_strided(t.detach(), param.shape, param.stride(), off)
                return view(self._m[i]), view(self._v[i])
        raise KeyError("FusedAdam.moments: the tensor is not a parameter of this optimizer's owners")

    @property
    def _step(self):
        """Step count of the first owner (the encoder): what the snapshot's `step` key has always meant."""
        return self._steps[0]

    # state: enough to resume (the reference never saved optimizer state, train_representation.py:123-130)
    def state_dict(self):
        return {"step": self._steps[0], "steps": list(self._steps), "exp_avg": [None if m is None else m.cpu() for m in self._m],
                "exp_avg_sq": [None if v is None else v.cpu() for v in self._v], "param_groups": [
                    {k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        steps = sd.get("steps")           # round-1 snapshots carry one shared `step`
        steps = [int(s) for s in steps] if steps is not None else [int(sd["step"])] * len(self.owners)
        n = len(self.owners)
        if len(steps) != n or len(sd["exp_avg"]) != n or len(sd["exp_avg_sq"]) != n:
            # e.g. saved with langweight=0 (encoder only) and resumed with a language head, or the reverse: silently
            # mis-assigning moments between owners would be worse than refusing
            raise ValueError(f"FusedAdam.load_state_dict: the snapshot holds optimizer state for {len(steps)} flat buffer(s) "
                             f"({len(sd['exp_avg'])} moment entries), this optimizer has {n} "
                             f"({', '.join(type(o).__name__ for o in self.owners)}); was it saved with a different langweight?")
        for i, o in enumerate(self.owners):
            m = sd["exp_avg"][i]
            if m is not None and m.numel() != o.flat_params().numel():
                raise ValueError(f"FusedAdam.load_state_dict: moments of owner {i} ({type(o).__name__}) have {m.numel()} elements, "
                                 f"its parameters {o.flat_params().numel()}")
        self._steps = steps
        for i, o in enumerate(self.owners):
            if sd["exp_avg"][i] is not None:
                dev = o.flat_params().device
                self._m[i] = sd["exp_avg"][i].to(dev)
                self._v[i] = sd["exp_avg_sq"][i].to(dev)
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD semantics (momentum, dampening, weight_decay, nesterov) as one HIP kernel per flat buffer. The reference
    only ever builds Adam (models_r3m.py:76); this is the plain alternative on the same owners protocol."""

    def __init__(self, owners, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        self.owners = list(owners)
        params = [p for o in self.owners for p in o.parameters()]
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov))
        self._steps = [0] * len(self.owners)
        self._buf = [None] * len(self.owners)
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none=True):
        for o in self.owners:
            o.mark_grads_stale()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FusedSGD.step(closure) is not supported")
        g0 = self.param_groups[0]
        L = _lib.lib()
        for i, o in enumerate(self.owners):
            p = o.flat_params()
            if not p.is_cuda:
                raise RuntimeError("r3m_amd.FusedSGD: parameters must live on the GPU (HIP kernel, no CPU fallback)")
            if not getattr(o, "has_grads", lambda: True)():
                continue
            g = o.flat_grads()
            if g0["momentum"] != 0 and (self._buf[i] is None or self._buf[i].device != p.device or self._buf[i].numel() != p.numel()):
                self._buf[i] = torch.zeros_like(p)
            buf_ptr = None if self._buf[i] is None else self._buf[i].data_ptr()
            self._steps[i] += 1
            with _lib.on(p):
                _lib.check(L.r3m_sgd_step(p.data_ptr(), g.data_ptr(), buf_ptr, p.numel(), float(g0["lr"]), float(g0["momentum"]),
                                          float(g0["dampening"]), float(g0["weight_decay"]), int(bool(g0["nesterov"])),
                                          self._steps[i], float(self.grad_scale), _lib.stream_ptr(p.device)), "sgd_step")

    def __getstate__(self):
        st = super().__getstate__()
        st.update(owners=self.owners, _steps=self._steps, _buf=self._buf, grad_scale=self.grad_scale)
        return st
