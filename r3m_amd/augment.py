"""On-GPU `rc` / `rctraj` augmentation: RandomResizedCrop(224, scale=(0.2, 1.0)) as used by the reference's loader
(/root/reference/r3m/utils/data_loaders.py:47-50,81-102). Box sampling restates torchvision's RandomResizedCrop.get_params
(third-party, un-vendored: up to 10 tries of area ~ U(scale)*A, log-uniform aspect in (3/4, 4/3), else a centre crop clamped
to the ratio range); the resample is the HIP kernel csrc/augment.hip."""
import math

import torch

from . import _lib


def _box_randoms(n, generator):
    """The random numbers of n boxes: 10 (area, log-ratio) tries each + one (top, left) position pair, float64 in [0,1)."""
    u = torch.rand((n, 22), dtype=torch.float64, generator=generator)
    return u[:, 0:10], u[:, 10:20], u[:, 20], u[:, 21]


def _fallback_box(height, width, ratio):
    in_ratio = width / height
    if in_ratio < ratio[0]:
        w, h = width, int(round(width / ratio[0]))
    elif in_ratio > ratio[1]:
        h, w = height, int(round(height * ratio[1]))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def sample_boxes(n, height, width, scale=(0.2, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), generator=None):
    """[n,4] int32 (top, left, h, w) for n boxes in a handful of array ops (host RNG, torch CPU generator): torchvision's
    RandomResizedCrop.get_params algorithm — up to 10 tries of area ~ U(scale) * A with a log-uniform aspect ratio, the first
    try that fits wins and gets a uniform position, else the centre fallback — evaluated for all boxes and tries at once
    (round 1 looped in Python with 2-4 scalar RNG calls per box: 6.5 % of the ResNet-34 bf16 step at 512 clips).
    The arithmetic runs in numpy on purpose: ~20 torch CPU ops per step woke torch's intra-op thread pool (one thread per core of
    a 256-thread host) next to the thread that launches the kernels — measured +7 ... +85 ms per 94 ms step when this ran after
    other workloads in the same process (bench.py's secondary configs[4]; 93.2 ms with torch.set_num_threads(1))."""
    import numpy as np
    area = float(height * width)
    ua, ur, ui, uj = (t.numpy() for t in _box_randoms(n, generator))
    target = area * (scale[0] + (scale[1] - scale[0]) * ua)
    ar = np.exp(math.log(ratio[0]) + (math.log(ratio[1]) - math.log(ratio[0])) * ur)
    w = np.round(np.sqrt(target * ar)).astype(np.int64)
    h = np.round(np.sqrt(target / ar)).astype(np.int64)
    ok = (w > 0) & (w <= width) & (h > 0) & (h <= height)
    first = np.argmax(ok, axis=1)                                   # first try that fits (0 when none does: masked below)
    any_ok = ok.any(axis=1)
    rows = np.arange(n)
    hh, ww = h[rows, first], w[rows, first]
    top = np.minimum(np.floor(ui * (height - hh + 1).astype(np.float64)).astype(np.int64), height - 1)
    left = np.minimum(np.floor(uj * (width - ww + 1).astype(np.float64)).astype(np.int64), width - 1)
    out = np.stack([top, left, hh, ww], axis=1)
    fb = np.asarray(_fallback_box(height, width, ratio), dtype=np.int64)
    out = np.where(any_ok[:, None], out, fb[None, :])
    return torch.from_numpy(out.astype(np.int32))


def _sample_boxes_scalar(n, height, width, scale=(0.2, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), generator=None):
    """The same algorithm box by box, try by try (the shape of torchvision's get_params loop) on the same random numbers:
    the checker of the vectorised form (tests/test_gpu_augment.py)."""
    area = height * width
    log_r = (math.log(ratio[0]), math.log(ratio[1]))
    ua, ur, ui, uj = _box_randoms(n, generator)
    out = torch.empty((n, 4), dtype=torch.int32)
    for b in range(n):
        box = None
        for t in range(10):
            target = area * (scale[0] + (scale[1] - scale[0]) * float(ua[b, t]))
            ar = math.exp(log_r[0] + (log_r[1] - log_r[0]) * float(ur[b, t]))
            w = int(round(math.sqrt(target * ar)))
            h = int(round(math.sqrt(target / ar)))
            if 0 < w <= width and 0 < h <= height:
                box = (min(int(math.floor(float(ui[b]) * (height - h + 1))), height - 1),
                       min(int(math.floor(float(uj[b]) * (width - w + 1))), width - 1), h, w)
                break
        if box is None:
            box = _fallback_box(height, width, ratio)
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
    boxes = _lib.upload_small(boxes, frames.device, torch.int32).contiguous()
    assert boxes.shape[0] * frames_per_box == N
    out = torch.empty((N, C, out_hw[0], out_hw[1]), dtype=torch.float32, device=frames.device)
    with _lib.on(frames):
        _lib.check(_lib.lib().r3m_crop_resize(frames.data_ptr(), 1 if frames.dtype == torch.uint8 else 0, boxes.data_ptr(),
                                              out.data_ptr(), N, C, H, W, out_hw[0], out_hw[1], frames_per_box,
                                              _lib.stream_ptr(frames.device)), "crop_resize")
    return out


class _NotATensor(AttributeError, TypeError):
    """Raised for tensor attributes a CroppedClips handle does not have. An AttributeError, so that hasattr() / getattr(x, name,
    default) probing by generic code (collate / transfer helpers, debuggers) keeps working; also a TypeError, which is what direct
    misuse of the handle as a tensor is."""


class CroppedClips:
    """Raw clips + crop boxes standing where the cropped frames [B,T,3,224,224] would stand. The encoder resamples the boxes
    inside its stem pre-pass (r3m_resnet_forward_crop: uint8 -> normalised stem image in one gather pass), so the cropped fp32
    frames — 0.6 MB each, written and read back in round 1 — never exist. Quacks enough like a tensor for Trainer.update
    (`.shape`, `.reshape(B*T, 3, 224, 224)`) and the prefetcher (`.float()`, `.record_stream()`); `.materialize()` gives the
    pixels (bit-identical to what the encoder sees)."""

    def __init__(self, raw, boxes, frames_per_box, out_hw=(224, 224)):
        if not raw.is_cuda:
            raise RuntimeError("r3m_amd.augment: HIP kernel needs a CUDA/HIP tensor (no CPU fallback)")
        if raw.dtype not in (torch.uint8, torch.float32):
            raw = raw.float()
        lead = raw.shape[:-3]
        self.raw = raw.reshape(-1, *raw.shape[-3:]).contiguous()                       # [N,3,H,W]
        self.boxes = _lib.upload_small(boxes, raw.device, torch.int32).contiguous()
        self.frames_per_box = int(frames_per_box)
        assert self.boxes.shape[0] * self.frames_per_box == self.raw.shape[0] and self.raw.shape[1] == 3
        self.out_hw = tuple(out_hw)
        self.shape = torch.Size((*lead, 3, *self.out_hw))
        self.device, self.dtype, self.is_cuda = raw.device, torch.float32, True

    def dim(self):
        return len(self.shape)

    def reshape(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        n = self.raw.shape[0]
        if -1 in shape:
            known = -math.prod(shape)
            shape = tuple(n * 3 * self.out_hw[0] * self.out_hw[1] // known if d == -1 else d for d in shape)
        if math.prod(shape) != math.prod(self.shape) or tuple(shape[-3:]) != (3, *self.out_hw):
            raise ValueError(f"CroppedClips.reshape{shape}: only the leading (clip, frame) dimensions can be regrouped")
        out = object.__new__(CroppedClips)
        out.__dict__.update(self.__dict__)
        out.shape = torch.Size(shape)
        return out

    def float(self):
        return self

    def contiguous(self):
        return self

    def record_stream(self, stream):
        self.raw.record_stream(stream)
        self.boxes.record_stream(stream)

    def materialize(self):
        return crop_resize(self.raw, self.boxes, self.frames_per_box, self.out_hw).reshape(self.shape)

    # Anything beyond the regrouping above needs pixels: these materialise (one stand-alone crop pass). The handle is NOT a
    # tensor — frame order is fixed by `raw`/`boxes`, so indexing or permuting returns plain tensors, never another handle.
    def __getitem__(self, idx):
        return self.materialize()[idx]

    def to(self, *args, **kwargs):
        dev = kwargs.get("device", args[0] if args and isinstance(args[0], (str, torch.device, int)) else None)
        dtype = kwargs.get("dtype", next((a for a in args if isinstance(a, torch.dtype)), None))
        if (dev is None or torch.device(dev) == self.device) and dtype in (None, torch.float32):
            return self
        return self.materialize().to(*args, **kwargs)

    def cpu(self):
        return self.materialize().cpu()

    def numpy(self):
        return self.cpu().numpy()

    def __getattr__(self, name):
        # only reached for attributes the handle does not have: fail with the contract instead of a bare AttributeError
        if name.startswith("__"):
            raise AttributeError(name)
        raise _NotATensor(f"CroppedClips.{name}: the fused rc/rctraj batch is a handle (raw clips + crop boxes), not a tensor; it supports "
                        f"shape / dim / reshape of the leading dims / float / contiguous / record_stream / to / cpu / indexing — call "
                        f".materialize() for the cropped [.., 3, 224, 224] float tensor, or pass fused=False to random_resized_crop")


def random_resized_crop(batch, per_clip=True, generator=None, fused=False):
    """batch [B,5,3,H,W] -> [B,5,3,224,224]; per_clip=True is `rctraj` (one box per clip), False is `rc`
    (/root/reference/r3m/utils/data_loaders.py:88-102). fused=True returns a CroppedClips handle instead of pixels: the crop
    then runs inside the encoder's stem pre-pass."""
    B, T = batch.shape[:2]
    H, W = batch.shape[-2:]
    boxes = sample_boxes(B if per_clip else B * T, H, W, generator=generator)
    if fused:
        return CroppedClips(batch, boxes, T if per_clip else 1)
    out = crop_resize(batch.reshape(B * T, *batch.shape[2:]), boxes, T if per_clip else 1)
    return out.reshape(B, T, *out.shape[1:])


def resize_center_crop_geometry(h, w, size=256, crop=224):
    """torchvision 0.8.2 geometry of transforms.Resize(size) (smaller edge -> size, the other int(size * long / short)) followed by
    CenterCrop(crop) (offsets int(round((dim - crop) / 2))) — un-vendored third-party code restated:
    returns (resized_h, resized_w, top, left)."""
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nh, nw = size, int(size * w / h)
    return nh, nw, int(round((nh - crop) / 2.0)), int(round((nw - crop) / 2.0))


def resize_center_crop(frames, size=256, crop=224):
    """frames [N,C,H,W] uint8 or float (0..255) on the GPU -> [N,C,crop,crop] float32 0..255: Resize(size) + CenterCrop(crop) of
    R3M.forward (/root/reference/r3m/models/models_r3m.py:85-90) as ONE HIP gather pass (csrc/augment.hip)."""
    if not frames.is_cuda:
        raise RuntimeError("r3m_amd.augment: HIP kernel needs a CUDA/HIP tensor (no CPU fallback)")
    if frames.dtype not in (torch.uint8, torch.float32):
        frames = frames.float()
    frames = frames.contiguous()
    N, C, H, W = frames.shape
    nh, nw, top, left = resize_center_crop_geometry(H, W, size, crop)
    if nh < crop or nw < crop:
        raise ValueError(f"resize_center_crop: {H}x{W} resizes to {nh}x{nw}, smaller than the {crop}x{crop} crop")
    out = torch.empty((N, C, crop, crop), dtype=torch.float32, device=frames.device)
    with _lib.on(frames):
        _lib.check(_lib.lib().r3m_resize_crop(frames.data_ptr(), 1 if frames.dtype == torch.uint8 else 0, out.data_ptr(), N, C, H, W,
                                              nh, nw, top, left, crop, crop, _lib.stream_ptr(frames.device)), "resize_crop")
    return out
