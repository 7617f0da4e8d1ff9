"""Thin autograd / tensor wrappers over the C ABI (include/r3m_hip.h). PyTorch here is plumbing: allocation, streams,
autograd bookkeeping. Every function requires CUDA (HIP) tensors and raises otherwise — no eager fallback.
"""
import torch

from . import _lib

METRIC_SLOTS = {"l2loss": 0, "l1loss": 1, "l0loss": 2, "tcnloss": 3, "aligned": 4, "rewloss": 5, "rewacc1": 6, "rewacc2": 7,
                "rewacc3": 8, "full_loss": 9}


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("r3m_amd.ops: HIP kernels need CUDA/HIP tensors (no CPU fallback)")


def inverse_permutations(perm):
    """perm [P,B] int -> inverse [P,B] (iperm[p, perm[p, i]] = i)."""
    P, B = perm.shape
    inv = torch.empty_like(perm)
    ar = torch.arange(B, dtype=perm.dtype, device=perm.device).expand(P, B)
    inv.scatter_(1, perm.long(), ar)
    return inv


class _R3MLossFn(torch.autograd.Function):
    """full_loss(alle [, scores]) with the reference's metrics as a side output; gradient computed in the same launches."""

    @staticmethod
    def forward(ctx, alle, perm, scores, mask, l2w, l1w, tcnw, langw, l2dist):
        _need_cuda(alle, perm, scores, mask)
        L = _lib.lib()
        B, five, D = alle.shape
        assert five == 5
        alle = alle.contiguous()
        dev = alle.device
        ws_bytes = L.r3m_loss_workspace_bytes(B)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        metrics = torch.zeros(16, dtype=torch.float32, device=dev)
        dalle = torch.empty_like(alle)
        if perm is None:
            perm = torch.arange(B, dtype=torch.int32, device=dev).repeat(6, 1)
        perm = perm.to(device=dev, dtype=torch.int32).contiguous()
        iperm = inverse_permutations(perm).contiguous()
        st = _lib.stream_ptr(dev)
        dscore = None
        have_lang = scores is not None and langw > 0
        with _lib.on(alle):
            _lib.check(L.r3m_loss_tcn_lp(alle.data_ptr(), perm.data_ptr(), iperm.data_ptr(), dalle.data_ptr(), ws.data_ptr(), ws_bytes,
                                         B, D, 1 if l2dist else 0, float(l2w), float(l1w), float(tcnw), st), "loss_tcn_lp")
            if have_lang:
                scores = scores.contiguous()
                mask = mask.to(device=dev, dtype=torch.float32).contiguous()
                dscore = torch.empty_like(scores)
                _lib.check(L.r3m_loss_lang_infonce(scores.data_ptr(), mask.data_ptr(), dscore.data_ptr(), ws.data_ptr(), ws_bytes, B,
                                                   float(langw), st), "loss_lang_infonce")
            _lib.check(L.r3m_loss_finalize(ws.data_ptr(), ws_bytes, B, 1 if have_lang else 0, metrics.data_ptr(), float(l2w),
                                           float(l1w), float(tcnw), float(langw) if have_lang else 0.0, st), "loss_finalize")
        ctx.save_for_backward(dalle, dscore)
        ctx.mark_non_differentiable(metrics)
        return metrics[9].clone(), metrics

    @staticmethod
    def backward(ctx, gfull, _gmetrics):
        dalle, dscore = ctx.saved_tensors
        ga = dalle * gfull
        gs = None if dscore is None else dscore * gfull
        return ga, None, gs, None, None, None, None, None, None


def r3m_loss(alle, perm, l2weight, l1weight, tcnweight, l2dist=True, scores=None, mask=None, langweight=0.0):
    """alle [B,5,D]; perm [6,B] TCN permutations in the reference's draw order; scores [15,B] + mask [B] for the language
    head (optional). Returns (full_loss 0-d tensor with grad, metrics[16] tensor — slots in METRIC_SLOTS)."""
    return _R3MLossFn.apply(alle, perm, scores, mask, l2weight, l1weight, tcnweight, langweight, l2dist)
