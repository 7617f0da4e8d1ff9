"""Language side of R3M on the HIP path — same classes as /root/reference/r3m/models/models_language.py:

  LangEncoder(device, finetune=False, scratch=False)   frozen DistilBERT sentence features, lang_size = 768   (:13-35)
  LanguageReward(ltype, im_dim, hidden_dim, lang_dim, simfunc=None)   MLP [2D+768 -> H -> H -> H -> H -> 1]    (:37-55)

What changes is the schedule, not the math: the reference calls get_reward 15 times per step and re-runs DistilBERT each
time on the same sentences (trainer.py:72-92). Here the sentence features are computed once per step and the 15 MLP
evaluations are one batched [15B, 2D+768] pass in csrc/lang.hip (`LanguageReward.batched_scores`).

DistilBERT itself is a frozen feature extractor outside the kernel scope: LangEncoder accepts
precomputed [B,768] features (BASELINE config 3: "frozen DistilBERT text feats"), a sentence->feature cache, or, when the
HuggingFace weights are on disk, runs the model once per batch with the reference's mean-over-all-positions pooling.
"""
import math

import torch
import torch.nn as nn

from . import _lib
from .ops import inverse_permutations

epsilon = 1e-8


class LangEncoder(nn.Module):
    """Frozen sentence features. `mask_padding=False` (default) is the reference: `last_hidden_state.mean(1)` over ALL token
    positions, padding included (models_language.py:34), so a sentence's feature depends on the longest sentence of its batch;
    `mask_padding=True` averages over the real tokens only (batch-independent — what an offline per-sentence cache assumes).
    Features are computed ONCE per call; Trainer.update calls this once per step where the reference re-runs DistilBERT in each
    of its 15 get_reward calls on the same sentences (trainer.py:72-92)."""

    def __init__(self, device, finetune=False, scratch=False, mask_padding=False):
        super().__init__()
        self.device = device
        self.modelname = "distilbert-base-uncased"
        self.lang_size = 768
        self.mask_padding = bool(mask_padding)
        self.feature_cache = {}       # sentence -> [768] tensor (offline-precomputed features)
        self._hf = None               # (tokenizer, model): use_backend(), or loaded lazily from local files only
        self.encoder_calls = 0        # transformer passes so far (tests: once per step, none on cache hits)

    def use_backend(self, tokenizer, model):
        """Plug a tokenizer + transformer (HuggingFace calling convention) instead of loading `distilbert-base-uncased` from
        the local cache. The model is frozen: eval mode, no gradients — the reference's DistilBERT is never trained either
        (under no_grad, models_language.py:29), but follows model.train() into dropout; here it does not."""
        model = model.to(self.device).eval()
        for p in model.parameters():
            p.requires_grad_(False)
        self._hf = (tokenizer, model)
        return self

    def train(self, mode=True):       # the frozen text model stays in eval mode whatever the step does (trainer.py:31)
        super().train(mode)
        if self._hf is not None:
            self._hf[1].eval()
        return self

    def encode(self, langs):
        """[len(langs), 768] features of a list of sentences in ONE transformer pass (models_language.py:29-34)."""
        tok, model = self._load_hf()
        with torch.no_grad():
            enc = tok(list(langs), return_tensors="pt", padding=True)
            am = enc["attention_mask"].to(self.device)
            out = model(enc["input_ids"].to(self.device), attention_mask=am).last_hidden_state
            self.encoder_calls += 1
            if self.mask_padding:
                w = am.to(out.dtype).unsqueeze(-1)
                return (out * w).sum(1) / w.sum(1).clamp_min(1.0)
            return out.mean(1)         # mean over ALL positions incl. padding, as the reference does

    def precompute(self, sentences, batch_size=256):
        """Fill feature_cache for a corpus (offline). With the reference's pooling a feature depends on its
        batch's longest sentence, so a cache is only faithful with mask_padding=True — enforced."""
        if not self.mask_padding:
            raise ValueError("LangEncoder.precompute: the reference's mean-over-padding features depend on the batch they were "
                             "computed in; build the cache with LangEncoder(..., mask_padding=True)")
        todo = [s for s in dict.fromkeys(sentences) if s not in self.feature_cache]
        for i in range(0, len(todo), batch_size):
            chunk = todo[i:i + batch_size]
            for s, f in zip(chunk, self.encode(chunk)):
                self.feature_cache[s] = f.detach().clone()
        return len(todo)

    def _load_hf(self):
        if self._hf is None:
            try:
                from transformers import AutoModel, AutoTokenizer
                tok = AutoTokenizer.from_pretrained(self.modelname, local_files_only=True)
                model = AutoModel.from_pretrained(self.modelname, local_files_only=True).to(self.device)
            except Exception as e:  # noqa: BLE001
                raise RuntimeError(
                    "r3m_amd.LangEncoder: sentence strings were passed but the DistilBERT weights "
                    f"({self.modelname}) are not available locally ({type(e).__name__}). Pass precomputed [B,768] "
                    "features instead of strings, or fill LangEncoder.feature_cache.") from e
            model.eval()
            self._hf = (tok, model)
        return self._hf

    def forward(self, langs):
        if torch.is_tensor(langs):                     # precomputed frozen features
            return langs
        try:
            langs = langs.tolist()
        except AttributeError:
            pass
        if langs and all(s in self.feature_cache for s in langs):
            return torch.stack([self.feature_cache[s] for s in langs]).to(self.device)
        return self.encode(langs)


class _Node(nn.Module):
    pass


class _BatchedRewardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alle, feats, perm, anchor, module):
        L = _lib.lib()
        B, _, D = alle.shape
        alle = alle.contiguous()
        feats = feats.to(dtype=torch.float32).contiguous()
        perm = perm.to(device=alle.device, dtype=torch.int32).contiguous()
        ws_bytes = L.r3m_langrew_workspace_bytes(B, D, module.hidden_dim, module.lang_dim)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=alle.device)
        scores = torch.empty((15, B), dtype=torch.float32, device=alle.device)
        with _lib.on(alle):
            dt = 1 if getattr(module, "precision", "fp32") == "bf16" else 0
            _lib.check(L.r3m_langrew_forward_dt(alle.data_ptr(), feats.data_ptr(), perm.data_ptr(), module.flat_params().data_ptr(),
                                                scores.data_ptr(), ws.data_ptr(), ws_bytes, B, D, module.hidden_dim, module.lang_dim,
                                                dt, _lib.stream_ptr(alle.device)), "langrew_forward")
        ctx.module, ctx.ws, ctx.ws_bytes, ctx.dims, ctx.dt = module, ws, ws_bytes, (B, D), dt
        ctx.save_for_backward(perm)
        return scores

    @staticmethod
    def backward(ctx, dscores):
        L = _lib.lib()
        (perm,) = ctx.saved_tensors
        module = ctx.module
        B, D = ctx.dims
        iperm = inverse_permutations(perm).contiguous()
        dalle = torch.zeros((B, 5, D), dtype=torch.float32, device=dscores.device)
        g = module.flat_grads()
        accumulate = 0 if module._grad_fresh else 1
        dscores = dscores.contiguous()
        with _lib.on(dscores):
            _lib.check(L.r3m_langrew_backward_dt(dscores.data_ptr(), iperm.data_ptr(), module.flat_params().data_ptr(),
                                                 g.data_ptr(), dalle.data_ptr(), ctx.ws.data_ptr(), ctx.ws_bytes, B, D, module.hidden_dim,
                                                 module.lang_dim, accumulate, ctx.dt, _lib.stream_ptr(dscores.device)), "langrew_backward")
        module._grad_fresh = False
        module._has_grads = True
        ctx.ws = None
        return dalle, None, None, None, None


class _RewardCallFn(torch.autograd.Function):
    """score = G(e0, eg, le) for ONE call, differentiable in e0, eg, le and the head's parameters — what the reference's
    trainer does 15 times per step through autograd (trainer.py:72-92). All arithmetic in csrc/lang.hip (r3m_langrew_call_*):
    concat -> 4 x (Linear + ReLU) on the MFMA gather-GEMM -> Linear(H -> 1) GEMV; backward = the batched pass's kernels on
    this call's rows. Parameter gradients accumulate into the module's flat gradient buffer across the calls of a step."""

    @staticmethod
    def forward(ctx, e0, eg, le, anchor, module):
        L = _lib.lib()
        R, D = e0.shape
        ws_bytes = L.r3m_langrew_call_workspace_bytes(R, D, module.hidden_dim, module.lang_dim)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=e0.device)
        score = torch.empty((R,), dtype=torch.float32, device=e0.device)
        with _lib.on(e0):
            _lib.check(L.r3m_langrew_call_forward(e0.data_ptr(), eg.data_ptr(), le.data_ptr(), module.flat_params().data_ptr(),
                                                  score.data_ptr(), ws.data_ptr(), ws_bytes, R, D, module.hidden_dim,
                                                  module.lang_dim, _lib.stream_ptr(e0.device)), "langrew_call_forward")
        ctx.module, ctx.ws, ctx.ws_bytes, ctx.dims = module, ws, ws_bytes, (R, D)
        return score

    @staticmethod
    def backward(ctx, dscore):
        L = _lib.lib()
        module = ctx.module
        R, D = ctx.dims
        if ctx.ws is None:
            raise RuntimeError("r3m_amd.LanguageReward: backward through the same get_reward call twice (activations were released)")
        dscore = dscore.to(torch.float32).contiguous()
        need = ctx.needs_input_grad
        dev = dscore.device
        de0 = torch.empty((R, D), dtype=torch.float32, device=dev) if need[0] else None
        deg = torch.empty((R, D), dtype=torch.float32, device=dev) if need[1] else None
        dle = torch.empty((R, module.lang_dim), dtype=torch.float32, device=dev) if need[2] else None
        g = module.flat_grads()
        accumulate = 0 if module._grad_fresh else 1
        with _lib.on(dscore):
            _lib.check(L.r3m_langrew_call_backward(dscore.data_ptr(), module.flat_params().data_ptr(), g.data_ptr(), _lib.ptr(de0),
                                                   _lib.ptr(deg), _lib.ptr(dle), ctx.ws.data_ptr(), ctx.ws_bytes, R, D,
                                                   module.hidden_dim, module.lang_dim, accumulate, _lib.stream_ptr(dev)),
                       "langrew_call_backward")
        module._grad_fresh = False
        module._has_grads = True
        ctx.ws = None
        return de0, deg, dle, None, None


class LanguageReward(nn.Module):
    def __init__(self, ltype, im_dim, hidden_dim, lang_dim, simfunc=None, precision="fp32"):
        super().__init__()
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"LanguageReward: precision {precision!r} (fp32 or bf16)")
        # "bf16": the batched training pass stores the MLP's activations bf16 and runs its Linears on the bf16 GEMM kernels (fp32
        # accumulation, fp32 master weights / gradients) — autocast(bfloat16) around the reference's get_reward calls, except that
        # a hidden activation is rounded twice (GEMM result, then bias + ReLU; csrc/lang.hip). The single-call form (forward /
        # R3M.get_reward) stays fp32: training-time and evaluation-time scores of a bf16 model differ at bf16 level.
        self.precision = precision
        self.ltype = ltype
        self.sim = simfunc
        self.sigm = nn.Sigmoid()
        self.im_dim, self.hidden_dim, self.lang_dim = im_dim, hidden_dim, lang_dim
        dims = [(hidden_dim, 2 * im_dim + lang_dim)] + [(hidden_dim, hidden_dim)] * 3 + [(1, hidden_dim)]
        n = sum(o * i + o for o, i in dims)
        self._n = n
        self._n_padded = (n + 3) // 4 * 4            # the fused Adam kernel works on float4
        self._layout = []                             # (name, offset, shape)
        flat = torch.zeros(self._n_padded, dtype=torch.float32)
        self.pred = _Node()
        off = 0
        for li, (o, i) in zip((0, 2, 4, 6, 8), dims):  # nn.Sequential indices of the Linear layers (models_language.py:43-51)
            node = _Node()
            self.pred.add_module(str(li), node)
            node.register_parameter("weight", nn.Parameter(flat[off:off + o * i].view(o, i)))
            self._layout.append((f"pred.{li}.weight", off, (o, i)))
            off += o * i
            node.register_parameter("bias", nn.Parameter(flat[off:off + o].view(o)))
            self._layout.append((f"pred.{li}.bias", off, (o,)))
            off += o
        self._flat_p = flat
        self._flat_g = None
        self._grad_fresh = True
        self._has_grads = False
        self.reset_parameters()

    def reset_parameters(self):
        """nn.Linear default init (kaiming_uniform(a=sqrt(5)) weight, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) bias)."""
        P = dict(self.named_parameters())
        with torch.no_grad():
            for name, off, shape in self._layout:
                if len(shape) == 2:
                    nn.init.kaiming_uniform_(P[name], a=math.sqrt(5))
                else:
                    fan_in = P[name.replace("bias", "weight")].shape[1]
                    bound = 1 / math.sqrt(fan_in)
                    nn.init.uniform_(P[name], -bound, bound)

    # ---- flat storage (same protocol as HipResNet) ----
    def _is_flat(self):
        P = dict(self.named_parameters())
        base, dev = self._flat_p.data_ptr(), self._flat_p.device
        return all(P[n].device == dev and P[n].data_ptr() == base + off * 4 for n, off, _ in self._layout)

    def _reflatten(self):
        P = dict(self.named_parameters())
        dev = next(iter(P.values())).device
        flat = torch.zeros(self._n_padded, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for name, off, shape in self._layout:
                v = flat[off:off + math.prod(shape)].view(shape)
                v.copy_(P[name])
                P[name].data = v
                P[name].grad = None
        self._flat_p, self._flat_g, self._grad_fresh, self._has_grads = flat, None, True, False

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self._reflatten()
        return self

    def flat_params(self):
        if not self._is_flat():
            self._reflatten()
        return self._flat_p

    def flat_grads(self):
        self.flat_params()
        if self._flat_g is None:
            self._flat_g = torch.zeros_like(self._flat_p)
            P = dict(self.named_parameters())
            for name, off, shape in self._layout:
                P[name].grad = self._flat_g[off:off + math.prod(shape)].view(shape)
        return self._flat_g

    def mark_grads_stale(self):
        """zero_grad(): the next backward overwrites the flat buffer, and until one arrives the head HAS no gradient —
        torch.optim.Adam skips parameters whose .grad is None after zero_grad(set_to_none=True), so must the fused step
        (and the data-parallel wrapper must not all-reduce a stale buffer)."""
        self._grad_fresh = True
        self._has_grads = False

    def has_grads(self):
        return self._has_grads

    # ---- compute ----
    def batched_scores(self, alle, feats, lang_perm):
        """All 15 reward evaluations of Trainer.update in one pass. alle [B,5,D] (requires grad), feats [B,lang_dim] frozen,
        lang_perm [9,B] = the reference's randperm draws (trainer.py:86-92). Returns scores [15,B] (rows: pos1-3, in-clip
        negatives 1-3, then for each k the permuted negatives of heads 1-3)."""
        if not alle.is_cuda:
            raise RuntimeError("r3m_amd.LanguageReward runs on the HIP path only (no CPU fallback)")
        anchor = self.pred._modules["0"].weight
        return _BatchedRewardFn.apply(alle, feats, lang_perm, anchor, self)

    def forward(self, e0, eg, le):
        """Single evaluation G(e0, eg, le) -> (score[B], {}), reference signature and semantics (models_language.py:53-55):
        differentiable in the embeddings and in the head's parameters, so the reference's own 15-call loop
        (trainer.py:72-92) trains the head and sends the language gradient into the encoder exactly as it does there.
        Trainer.update of this package uses batched_scores() instead (one pass for the 15 calls)."""
        if not e0.is_cuda:
            raise RuntimeError("r3m_amd.LanguageReward runs on the HIP path only (no CPU fallback)")
        lead = e0.shape[:-1]
        D = e0.shape[-1]
        if D != self.im_dim or eg.shape != e0.shape or le.shape[:-1] != lead or le.shape[-1] != self.lang_dim:
            raise ValueError(f"LanguageReward: e0 {tuple(e0.shape)}, eg {tuple(eg.shape)}, le {tuple(le.shape)} "
                             f"(expected [...,{self.im_dim}] x2 and [...,{self.lang_dim}])")
        e0 = e0.reshape(-1, D).to(torch.float32).contiguous()
        eg = eg.reshape(-1, D).to(torch.float32).contiguous()
        le = le.to(e0.device).reshape(-1, self.lang_dim).to(torch.float32).contiguous()
        anchor = self.pred._modules["0"].weight
        score = _RewardCallFn.apply(e0, eg, le, anchor, self)
        return score.reshape(*lead, 1).squeeze(), {}      # `.squeeze()` as the reference: B = 1 collapses to 0-d
