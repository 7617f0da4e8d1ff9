import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import r3m_ref
from r3m_amd import ops
from r3m_amd.models_language import LanguageReward
torch.manual_seed(0)
B, D = 256, 2048
g = torch.Generator().manual_seed(21)
alle_c = torch.rand((B, 5, D), generator=g) * 1.5
feats = (torch.randn((B, 768), generator=g) * 0.3)
mask = torch.ones(B); mask[::7] = 0.0
lang_perm = torch.stack([torch.randperm(B, generator=g) for _ in range(9)])
ref = r3m_ref.R3MRef(size=50, l2weight=0.0, l1weight=0.0, langweight=1.0, tcnweight=0.0, l2dist=True)
sd = {k: v.clone() for k, v in ref.lang_rew.state_dict().items()}
out = {}
for dt in (torch.float32, torch.float64):
    ref.lang_rew.to(dt)
    a_ref = alle_c.to(dt).clone().requires_grad_(True)
    fl, met, sc = r3m_ref.r3m_loss_ref(ref, a_ref, tcn_perm=None, lang_feats=feats.to(dt), lang_mask=mask.to(dt), lang_perm=lang_perm)
    ref.zero_grad(); fl.backward()
    out[dt] = (a_ref.grad.double().clone(), None, sc.detach().double().clone())
rew = LanguageReward(None, D, 1024, 768); rew.load_state_dict(sd); rew = rew.to("cuda")
alle = alle_c.to("cuda").requires_grad_(True)
scores = rew.batched_scores(alle, feats.to("cuda"), lang_perm.to(torch.int32).to("cuda"))
scores.retain_grad()
full, m = ops.r3m_loss(alle, None, 0.0, 0.0, 0.0, l2dist=True, scores=scores, mask=mask.to("cuda"), langweight=1.0)
rew.mark_grads_stale(); full.backward()
gh, dsh, sch = alle.grad.cpu().double(), scores.grad.cpu().double(), scores.detach().cpu().double()
(g32, ds32, sc32), (g64, ds64, sc64) = out[torch.float32], out[torch.float64]
def r(a, b): return float((a - b).abs().max() / b.abs().max())
print("scores : hip~64 %.3e cpu32~64 %.3e" % (r(sch, sc64), r(sc32, sc64)))
print("dalle  : hip~64 %.3e cpu32~64 %.3e  max %.3e" % (r(gh, g64), r(g32, g64), float(g64.abs().max())))
for slot in range(5):
    print(" slot", slot, "hip~64 %.3e cpu32~64 %.3e max %.3e" % (float((gh[:, slot] - g64[:, slot]).abs().max()), float((g32[:, slot] - g64[:, slot]).abs().max()), float(g64[:, slot].abs().max())))
for slot in (0, 1, 3):
    err = (gh[:, slot] - g64[:, slot]).abs().max(dim=1).values
    top = torch.argsort(err, descending=True)[:8]
    print("slot", slot, "top clips by error:", [(int(i), "%.2e" % float(err[i]), int(mask[i])) for i in top], " clips with err>1e-10:", int((err > 1e-10).sum()))
# relative error per element where it is largest
e0 = (gh[:, 0] - g64[:, 0])
i = int(e0.abs().max(dim=1).values.argmax()); j = int(e0[i].abs().argmax())
print("worst element clip", i, "dim", j, "hip", float(gh[i, 0, j]), "ref64", float(g64[i, 0, j]), "cpu32", float(g32[i, 0, j]))
print("is clip a fixed point / target of perms:", [(k, int(lang_perm[k, i]), int((lang_perm[k] == i).nonzero()[0])) for k in range(9)])
