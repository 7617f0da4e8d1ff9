"""Deterministic, platform-independent tensor generator (integer hash of (name, flat index) -> float), so that golden
fixtures can be regenerated bit-identically on the GPU box without shipping 45-94 MB of weights. Test infrastructure."""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def unit(name, n):
    """n floats in [0,1), float64, function of (name, index) only."""
    seed = np.uint64((zlib.crc32(name.encode("utf-8")) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + seed
        h = _splitmix(_splitmix(idx))
    return (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def uniform(name, shape, lo, hi):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * unit(name, n)).astype(np.float32).reshape(shape)


def frames(name, shape):
    """uint8-valued frames as float32 in 0..255 (what the reference's loader yields, data_loaders.py:98-104)."""
    n = int(np.prod(shape))
    return np.floor(unit(name, n) * 256.0).astype(np.float32).reshape(shape)


def permutation(name, n):
    return np.argsort(unit(name, n), kind="stable").astype(np.int64)


def resnet_state_dict(named_shapes, tag="w"):
    """Kaiming-scaled uniform conv weights, non-trivial BatchNorm affine and running statistics.
    named_shapes: iterable of (key, shape) in state-dict order (keys as in torchvision, no prefix)."""
    out = {}
    for key, shape in named_shapes:
        if key.endswith("num_batches_tracked"):
            out[key] = np.zeros((), dtype=np.int64)
        elif key.endswith("running_mean"):
            out[key] = uniform(tag + key, shape, -0.1, 0.1)
        elif key.endswith("running_var"):
            out[key] = uniform(tag + key, shape, 0.5, 1.5)
        elif len(shape) == 4:
            fan_out = shape[0] * shape[2] * shape[3]
            a = float(np.sqrt(3.0) * np.sqrt(2.0 / fan_out))
            out[key] = uniform(tag + key, shape, -a, a)
        elif key.endswith(".weight"):
            out[key] = uniform(tag + key, shape, 0.5, 1.5)
        else:
            out[key] = uniform(tag + key, shape, -0.2, 0.2)
    return out


def smooth_frames(name, shape, grid=7):
    """uint8-valued LOW-FREQUENCY frames [N,3,H,W] (float32 0..255): a coarse random colour grid per frame, bilinearly
    upsampled (half-pixel centres) and floored. On i.i.d.-noise frames every frame looks alike to a deep network and the
    bf16-activation arithmetic is ill-conditioned (tests/test_gpu_bf16.py); these frames differ from each other at the
    scales the network sees."""
    n, c, h, w = shape
    coarse = uniform(name, (n, c, grid, grid), 0.0, 255.0).astype(np.float64)

    def axis(size):
        src = np.clip((np.arange(size) + 0.5) * (grid / size) - 0.5, 0.0, None)
        i0 = np.minimum(src.astype(np.int64), grid - 1)
        i1 = np.minimum(i0 + 1, grid - 1)
        return i0, i1, src - i0

    y0, y1, fy = axis(h)
    x0, x1, fx = axis(w)
    rows = coarse[:, :, y0, :] * (1.0 - fy)[None, None, :, None] + coarse[:, :, y1, :] * fy[None, None, :, None]
    img = rows[:, :, :, x0] * (1.0 - fx) + rows[:, :, :, x1] * fx
    return np.clip(np.floor(img), 0, 255).astype(np.float32)


def resnet_state_dict_small_residual(named_shapes, size, scale=0.1, tag="w"):
    """resnet_state_dict with the gamma of every block's LAST BatchNorm scaled by `scale` (small residual branches, as in a
    zero-init-residual / trained network): the state in which the bf16-activation ResNet-50 is well conditioned in BOTH
    BatchNorm modes (tools/experiments/bf16_conditioning.py: gradient cosine bf16-vs-exact 0.997 with batch statistics,
    against 0.10-0.38 for the plain states)."""
    out = resnet_state_dict(named_shapes, tag)
    last = "bn3.weight" if size == 50 else "bn2.weight"
    for key in out:
        if key.startswith("layer") and key.endswith(last):
            out[key] = (out[key] * np.float32(scale)).astype(np.float32)
    return out


def resnet_state_dict_no_kink(named_shapes, size, tag="nk", shift=4.0):
    """resnet_state_dict under its own tag with the bias of the network's LAST BatchNorm shifted up by `shift`: its input is
    batch-normalised (~N(0,1) per channel), so the share of last-block pre-activations z = bn(y) + identity that lie near zero
    falls like the normal tail and — for the (tag, shift) found by tools/experiments/find_nokink.py — NO element of the float64
    forward lies within 1e-3 of zero on the 8 golden frames. No fp32 forward can then decide a ReLU of the last block
    differently from float64, and the gradient gate of the kink-free golden case needs no flip accounting.
    The ReLU is still active (a fraction of z is negative), the rest of the network is the plain generator."""
    out = resnet_state_dict(named_shapes, tag)
    last = ("layer4.2.bn3" if size == 50 else ("layer4.1.bn2" if size == 18 else "layer4.2.bn2")) + ".bias"
    out[last] = (out[last] + np.float32(shift)).astype(np.float32)
    return out


# The kink-free golden cases (G8): (weight tag, last-BatchNorm bias shift, frames tag) per encoder size — three independent draws each
# (tools/experiments/find_nokink.py found them: no float64 last-block pre-activation within 1e-3 of zero). State 0 is round 4's case.
NOKINK_STATES = {
    18: [("nk2", 4.0, "frames8nk"), ("nk6", 4.0, "frames8nke"), ("nk8", 4.0, "frames8nkg")],
    34: [("nk2", 4.0, "frames8nk"), ("nk3", 4.0, "frames8nkb"), ("nk8", 4.0, "frames8nkg")],
    50: [("nk2", 4.0, "frames8nk"), ("nk3", 4.0, "frames8nkb"), ("nk8", 4.0, "frames8nkg")],
}
NOKINK_SUFFIX = ["", "_b", "_c"]      # tests/golden/encoder_r{size}_nokink{suffix}.npz
