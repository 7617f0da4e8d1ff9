import numpy as np
import torch


def rel_err(a, b):
    """max|a-b| / max|b| and ||a-b||2/||b||2 (float64)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = a - b
    return float(np.abs(d).max() / max(np.abs(b).max(), 1e-30)), float(np.linalg.norm(d) / max(np.linalg.norm(b), 1e-30))


def rnd(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo


def nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def nchw(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).contiguous()
