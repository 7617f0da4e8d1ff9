import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import r3m_ref
from r3m_amd import ops
from r3m_amd.models_language import LanguageReward
B, D = 256, 2048
g = torch.Generator().manual_seed(21)
alle_c = torch.rand((B, 5, D), generator=g) * 1.5
feats = (torch.randn((B, 768), generator=g) * 0.3)
mask = torch.ones(B); mask[::7] = 0.0
lang_perm = torch.stack([torch.randperm(B, generator=g) for _ in range(9)])
tcn_perm = torch.stack([torch.randperm(B, generator=g) for _ in range(6)])
for l2dist in (True, False):
    ref = r3m_ref.R3MRef(size=50, l2weight=1e-5, l1weight=1e-5, langweight=1.0, tcnweight=1.0, l2dist=l2dist)
    sd = ref.lang_rew.state_dict()
    out = {}
    for dt in (torch.float32, torch.float64):
        ref.lang_rew.to(dt)
        a_ref = alle_c.to(dt).clone().requires_grad_(True)
        fl, met, _ = r3m_ref.r3m_loss_ref(ref, a_ref, tcn_perm=tcn_perm, lang_feats=feats.to(dt), lang_mask=mask.to(dt), lang_perm=lang_perm)
        ref.zero_grad(); fl.backward()
        out[dt] = a_ref.grad.clone().double()
    ref.lang_rew.to(torch.float32)
    rew = LanguageReward(None, D, 1024, 768); rew.load_state_dict({k: v.float() for k, v in sd.items()}); rew = rew.to("cuda")
    alle = alle_c.to("cuda").requires_grad_(True)
    scores = rew.batched_scores(alle, feats.to("cuda"), lang_perm.to(torch.int32).to("cuda"))
    full, m = ops.r3m_loss(alle, tcn_perm.to(torch.int32).to("cuda"), 1e-5, 1e-5, 1.0, l2dist=l2dist, scores=scores, mask=mask.to("cuda"), langweight=1.0)
    rew.mark_grads_stale(); full.backward()
    gh = alle.grad.cpu().double()
    g32, g64 = out[torch.float32], out[torch.float64]
    mx = float(g64.abs().max())
    print(f"l2dist={l2dist}: max|g64| {mx:.3e}  hip~fp64 {float((gh-g64).abs().max())/mx:.3e}  cpu32~fp64 {float((g32-g64).abs().max())/mx:.3e}  hip~cpu32 {float((gh-g32).abs().max())/mx:.3e}")
