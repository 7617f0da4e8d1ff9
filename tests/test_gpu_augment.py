"""-m gpu: rc / rctraj crop-resize kernel vs torch CPU (crop -> /255 -> F.interpolate bilinear -> *255), boxes given explicitly
('pin with a box-parameterised golden')."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32])
@pytest.mark.parametrize("per_clip", [True, False])
def test_crop_resize_matches_torch(hip, dtype, per_clip):
    from r3m_amd import augment
    g = torch.Generator().manual_seed(9)
    B, T, H, W = 3, 5, 256, 341
    frames = torch.randint(0, 256, (B, T, 3, H, W), generator=g, dtype=torch.uint8)
    nb = B if per_clip else B * T
    boxes = augment.sample_boxes(nb, H, W, generator=g)
    boxes[0] = torch.tensor([0, 0, H, W], dtype=torch.int32)          # whole frame
    boxes[1] = torch.tensor([H - 7, W - 5, 7, 5], dtype=torch.int32)  # tiny box in the corner (up-sampling, edge clamps)
    x = frames.reshape(B * T, 3, H, W)
    out = augment.crop_resize(x.to("cuda:0").to(dtype), boxes, T if per_clip else 1).cpu()
    fpb = T if per_clip else 1
    for n in range(B * T):
        t, l, h, w = [int(v) for v in boxes[n // fpb]]
        ref = F.interpolate(x[n:n + 1, :, t:t + h, l:l + w].float() / 255.0, size=(224, 224), mode="bilinear", align_corners=False) * 255.0
        torch.testing.assert_close(out[n:n + 1], ref, rtol=1e-5, atol=2e-4)
    assert out.min() >= 0 and out.max() <= 255.0 + 1e-3


def test_box_sampler_statistics():
    from r3m_amd import augment
    g = torch.Generator().manual_seed(1)
    b = augment.sample_boxes(400, 224, 224, generator=g).float()
    area = b[:, 2] * b[:, 3] / (224.0 * 224.0)
    ratio = b[:, 3] / b[:, 2]
    assert area.min() >= 0.19 and area.max() <= 1.0 and 0.5 < float(area.mean()) < 0.7
    assert ratio.min() > 0.70 and ratio.max() < 1.40
    assert (b[:, 0] + b[:, 2] <= 224).all() and (b[:, 1] + b[:, 3] <= 224).all() and (b[:, :2] >= 0).all()
