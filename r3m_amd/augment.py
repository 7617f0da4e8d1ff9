"""On-GPU `rc` / `rctraj` augmentation: RandomResizedCrop(224, scale=(0.2, 1.0)) as used by the reference's loader
(/root/reference/r3m/utils/data_loaders.py:47-50,81-102). Box sampling restates torchvision's RandomResizedCrop.get_params
(third-party, un-vendored: up to 10 tries of area ~ U(scale)*A, log-uniform aspect in (3/4, 4/3), else a centre crop clamped
to the ratio range); the resample is the HIP kernel csrc/augment.hip."""
import math

import torch

from . import _lib


def sample_boxes(n, height, width, scale=(0.2, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), generator=None):
    """[n,4] int32 (top, left, h, w), host RNG (torch CPU generator)."""
    area = height * width
    log_r = (math.log(ratio[0]), math.log(ratio[1]))
    out = torch.empty((n, 4), dtype=torch.int32)
    for b in range(n):
        box = None
        for _ in range(10):
            target = area * float(torch.empty(1).uniform_(scale[0], scale[1], generator=generator))
            ar = math.exp(float(torch.empty(1).uniform_(log_r[0], log_r[1], generator=generator)))
            w = int(round(math.sqrt(target * ar)))
            h = int(round(math.sqrt(target / ar)))
            if 0 < w <= width and 0 < h <= height:
                i = int(torch.randint(0, height - h + 1, (1,), generator=generator))
                j = int(torch.randint(0, width - w + 1, (1,), generator=generator))
                box = (i, j, h, w)
                break
        if box is None:
            in_ratio = width / height
            if in_ratio < ratio[0]:
                w, h = width, int(round(width / ratio[0]))
            elif in_ratio > ratio[1]:
                h, w = height, int(round(height * ratio[1]))
            else:
                w, h = width, height
            box = ((height - h) // 2, (width - w) // 2, h, w)
        out[b] = torch.tensor(box, dtype=torch.int32)
    return out


def crop_resize(frames, boxes, frames_per_box, out_hw=(224, 224)):
    """frames [N,C,H,W] uint8 or float32 (0..255) on the GPU, boxes [N/frames_per_box, 4] -> [N,C,224,224] float32."""
    if not frames.is_cuda:
        raise RuntimeError("r3m_amd.augment: HIP kernel needs a CUDA/HIP tensor (no CPU fallback)")
    if frames.dtype not in (torch.uint8, torch.float32):
        frames = frames.float()
    frames = frames.contiguous()
    N, C, H, W = frames.shape
    boxes = boxes.to(device=frames.device, dtype=torch.int32).contiguous()
    assert boxes.shape[0] * frames_per_box == N
    out = torch.empty((N, C, out_hw[0], out_hw[1]), dtype=torch.float32, device=frames.device)
    with _lib.on(frames):
        _lib.check(_lib.lib().r3m_crop_resize(frames.data_ptr(), 1 if frames.dtype == torch.uint8 else 0, boxes.data_ptr(),
                                              out.data_ptr(), N, C, H, W, out_hw[0], out_hw[1], frames_per_box,
                                              _lib.stream_ptr(frames.device)), "crop_resize")
    return out


def random_resized_crop(batch, per_clip=True, generator=None):
    """batch [B,5,3,H,W] -> [B,5,3,224,224]; per_clip=True is `rctraj` (one box per clip), False is `rc`."""
    B, T = batch.shape[:2]
    H, W = batch.shape[-2:]
    boxes = sample_boxes(B if per_clip else B * T, H, W, generator=generator)
    out = crop_resize(batch.reshape(B * T, *batch.shape[2:]), boxes, T if per_clip else 1)
    return out.reshape(B, T, *out.shape[1:])
