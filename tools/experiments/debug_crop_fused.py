"""GPU: where do the fused crop->stem pre-pass and crop -> stem_prep differ?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from r3m_amd import _lib, augment
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B, T, H, W = 3, 5, 256, 320
g = torch.Generator().manual_seed(21)
raw = torch.randint(0, 256, (B * T, 3, H, W), generator=g, dtype=torch.uint8).cuda()
boxes = augment.sample_boxes(B, H, W, generator=g)
boxes[0] = torch.tensor([H - 9, W - 6, 9, 6], dtype=torch.int32)
bd = boxes.cuda()
F = B * T
pix = augment.crop_resize(raw, boxes, T)
xa = torch.empty((F, 224, 224, 3), device="cuda")
xb = torch.full((F, 224, 224, 3), float("nan"), device="cuda")
assert L.r3m_stem_prep(pix.data_ptr(), xa.data_ptr(), F, st) == 0
assert L.r3m_stem_prep_crop(raw.data_ptr(), 1, bd.data_ptr(), T, H, W, xb.data_ptr(), F, 0, st) == 0
torch.cuda.synchronize()
d = (xa - xb).abs()
print("fp32 xn: mismatching elements", int((xa != xb).sum()), "of", xa.numel(), "max abs diff", float(d.max()), "nan", int(torch.isnan(xb).sum()))
bad = (xa != xb).nonzero()[:10]
for i in bad.tolist():
    print(i, float(xa[tuple(i)]), float(xb[tuple(i)]))
# per-frame mismatch counts
print("per frame:", [(int((xa[f] != xb[f]).sum())) for f in range(F)])
