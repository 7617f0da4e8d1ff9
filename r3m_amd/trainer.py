"""Trainer.update — one pre-training (or eval) step; same signature, metric keys and RNG consumption order as the reference
(/root/reference/r3m/trainer.py:21-162), restructured for the GPU:

  * the 5B frames go through the HIP encoder engine in one call (trainer.py:39-41);
  * LP norms + TCN InfoNCE + language InfoNCE are the fused loss kernels of csrc/loss.hip, forward and backward together
    (trainer.py:52-59, 64-118, 122-150) — the permutations are still drawn with torch.randperm on the CPU generator in the
    reference's order (language: (a,b,c) x 3 at :86-92, then TCN: (es0, es2) x 3 at :136-137), uploaded once;
  * the 15 reward-head evaluations are one batched [15B, 2D+768] GEMM chain (models_language.LanguageReward.batched);
    the frozen sentence features are computed once per step instead of 15 times (trainer.py:72-92 -> models_r3m.py:78-81);
  * with a data-parallel wrapper built with global_negatives=True the rank's [B,5,D] embeddings (and sentence features / mask)
    are all-gathered first and everything below runs on the global batch, as the reference's DataParallel step does on GPU 0
    (trainer.py:41: `alles` is the gathered output); permutations are drawn by every rank (same RNG consumption) and rank 0's
    are used everywhere;
  * all metrics come back in ONE device->host copy instead of ~10 .item() syncs, queued before the backward pass so that the
    host does not wait for the optimizer step (the next step is queued behind it).
"""
import time

import torch

from . import _lib, ops

epsilon = 1e-8


class Trainer:
    def __init__(self, eval_freq):
        self.eval_freq = eval_freq
        self._mh = None          # pinned host copy of the step's metrics

    def update(self, model, batch, step, eval=False):
        t0 = time.time()
        metrics = dict()
        if eval:
            model.eval()
        else:
            model.train()
        t1 = time.time()
        b_im, b_lang = batch
        t2 = time.time()
        core = model.module

        bs = b_im.shape[0]
        b_im_r = b_im.reshape(bs * 5, 3, 224, 224)
        alles = model(b_im_r)
        alle = alles.reshape(bs, 5, -1)
        gneg = bool(getattr(model, "global_negatives", False))
        if gneg:
            alle = model.gather(alle)                # [world * bs, 5, D], differentiable: this rank backpropagates its own rows
            bs = alle.shape[0]
        t3 = time.time()

        # ---- permutations, in the reference's order of torch.randperm draws ----
        scores = mask = None
        lang_perm = None
        if core.langweight > 0:
            lang_perm = torch.stack([torch.randperm(bs) for _ in range(3 * core.num_negatives)])  # (a,b,c) x num_neg
        tcn_perm = None
        if core.tcnweight > 0:
            tcn_perm = torch.stack([torch.randperm(bs) for _ in range(2 * core.num_negatives)]).to(torch.int32)
            tcn_perm = _lib.upload_small(tcn_perm, alle.device)      # pinned staging: a pageable copy would stall the host here
            if gneg:
                tcn_perm = model.share(tcn_perm)

        if core.langweight > 0:
            # b_lang: list[str] as in the reference, or precomputed frozen features [B,768] / (features, mask[B])
            lang_mask = None
            if isinstance(b_lang, (tuple, list)) and len(b_lang) == 2 and torch.is_tensor(b_lang[0]):
                b_lang, lang_mask = b_lang
            feats = core.lang_enc(b_lang).to(alle.device)
            lang_perm = _lib.upload_small(lang_perm, alle.device, torch.int32)
            if lang_mask is not None:
                mask = lang_mask.to(device=alle.device, dtype=torch.float32)
            elif torch.is_tensor(b_lang):
                mask = torch.ones(feats.shape[0], dtype=torch.float32, device=alle.device)
            else:   # videos without language are masked out (trainer.py:107-109)
                mask = torch.tensor([1.0 * (b != "") for b in b_lang], dtype=torch.float32, device=alle.device)
            if gneg:
                feats, mask, lang_perm = model.gather(feats), model.gather(mask), model.share(lang_perm)
            scores = core.lang_rew.batched_scores(alle, feats, lang_perm)
        t5 = time.time()

        full_loss, m = ops.r3m_loss(alle, tcn_perm, core.l2weight, core.l1weight, core.tcnweight, l2dist=core.l2dist,
                                    scores=scores, mask=mask, langweight=core.langweight)
        t6 = time.time()
        # The metrics are final once the objective's forward has run: their device->host copy is queued HERE, in front of the
        # backward pass, and the host waits for that copy only — it returns while the GPU still works through backward + Adam and
        # queues the next step behind them (stream order keeps the weights consistent). Waiting for the whole step instead left
        # the GPU idle for ~0.5 ms per step between Adam and the next step's first kernel (rocprofv3 trace, bf16 ResNet-50).
        ready = None
        if m.is_cuda:
            if getattr(self, "_mh", None) is None or self._mh.numel() != m.numel():
                self._mh = torch.empty(m.numel(), dtype=m.dtype, pin_memory=True)
            self._mh.copy_(m.detach().reshape(-1), non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
        if not eval:
            core.encoder_opt.zero_grad()
            full_loss.backward()
            sync = getattr(model, "finish_gradient_sync", None)
            if sync is not None:
                sync()
            core.encoder_opt.step()

        if ready is not None:
            ready.synchronize()          # the step's single host wait
            mh = self._mh.tolist()
        else:
            mh = m.tolist()
        for k in ("l2loss", "l1loss", "l0loss"):
            metrics[k] = mh[ops.METRIC_SLOTS[k]]
        if core.langweight > 0:
            for k in ("rewloss", "rewacc1", "rewacc2", "rewacc3"):
                metrics[k] = mh[ops.METRIC_SLOTS[k]]
        if core.tcnweight > 0:
            for k in ("tcnloss", "aligned"):
                metrics[k] = mh[ops.METRIC_SLOTS[k]]
        metrics["full_loss"] = mh[ops.METRIC_SLOTS["full_loss"]]
        t7 = time.time()
        st = (f"Load time {t1-t0}, Batch time {t2-t1}, Encode time {t3-t2}, Lang time {t5-t3}, "
              f"Loss time {t6-t5}, Backprop time {t7-t6}")
        return metrics, st
