"""not gpu, SELF-ARMING: runs only where `import torchvision` succeeds (it is not installed in the authoring image nor on the GPU
boxes so far — every test here then skips). torchvision is the reference's un-vendored dependency for three things this build had
to restate (VERDICT r4 "missing #5"); wherever the wheel exists these tests pin the restatements against it:

  * the ResNet-18 / 34 / 50 graph          reference call site /root/reference/r3m/models/models_r3m.py:46-52   <-> oracle/resnet_ref.py
  * RandomResizedCrop.get_params           /root/reference/r3m/utils/data_loaders.py:47-50                      <-> r3m_amd/augment.py
  * Resize(256) + CenterCrop(224)          /root/reference/r3m/models/models_r3m.py:87-89                      <-> oracle/r3m_ref.py
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

tv = pytest.importorskip("torchvision", reason="torchvision is not installed: the restatements stay pinned only by construction")


@pytest.mark.parametrize("size", [18, 34, 50])
def test_resnet_restatement_matches_torchvision(size):
    """Same state-dict key set (without fc: R3M replaces it by Identity, models_r3m.py:62), and the same float64 forward in train
    and eval mode (+ the same running statistics afterwards) on two frames under one deterministic state."""
    from oracle import detgen, resnet_ref
    ours = getattr(resnet_ref, f"resnet{size}")()
    ctor = getattr(tv.models, f"resnet{size}")
    try:
        theirs = ctor(weights=None)
    except TypeError:                                     # torchvision < 0.13 (the reference pins 0.8.2)
        theirs = ctor(pretrained=False)
    theirs.fc = torch.nn.Identity()
    ours.fc = torch.nn.Identity()
    k_ours = [k for k in ours.state_dict() if not k.startswith("fc.")]
    k_theirs = [k for k in theirs.state_dict() if not k.startswith("fc.")]
    assert k_ours == k_theirs
    shapes = [(k, tuple(v.shape)) for k, v in theirs.state_dict().items()]
    assert shapes == [(k, tuple(v.shape)) for k, v in ours.state_dict().items()]
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict(shapes, "tvpin").items()}
    ours.load_state_dict(sd)
    theirs.load_state_dict(sd)
    ours, theirs = ours.double(), theirs.double()
    x = torch.from_numpy(detgen.frames("tvpin_frames", (2, 3, 224, 224))).double() / 255.0
    for mode in ("eval", "train"):
        getattr(ours, mode)()
        getattr(theirs, mode)()
        with torch.no_grad():
            a, b = ours(x), theirs(x)
        assert a.shape == b.shape == (2, 512 if size < 50 else 2048)
        torch.testing.assert_close(a, b, rtol=1e-10, atol=1e-12)
    for (ka, va), (kb, vb) in zip(ours.state_dict().items(), theirs.state_dict().items()):
        assert ka == kb
        torch.testing.assert_close(va.double(), vb.double(), rtol=1e-10, atol=1e-12)


def test_box_sampler_matches_random_resized_crop_get_params(monkeypatch):
    """RandomResizedCrop.get_params draws its randoms one by one from torch's global generator (uniform_ for area and log-ratio of
    each try, randint for the position); r3m_amd.augment.sample_boxes draws 22 uniforms per box up front. Same ALGORITHM is what can
    be pinned: torchvision's draws are answered from the very uniforms sample_boxes consumed (uniform_(a, b) -> a + (b - a) u,
    randint(0, n) -> floor(u n)), then every box must agree."""
    from r3m_amd import augment
    T = tv.transforms
    scale, ratio = (0.2, 1.0), (3.0 / 4.0, 4.0 / 3.0)
    for (H, W) in ((256, 256), (224, 224), (200, 320), (97, 131), (31, 5)):
        n = 64
        g = torch.Generator().manual_seed(1234 + H)
        boxes = augment.sample_boxes(n, H, W, scale, ratio, generator=g)
        g = torch.Generator().manual_seed(1234 + H)
        ua, ur, ui, uj = augment._box_randoms(n, g)
        for b in range(n):
            feed = {"try": 0, "pos": 0, "phase": 0}

            def fake_uniform(self, lo=0.0, hi=1.0, *a, **k):
                t = feed["try"]
                u = float(ua[b, t]) if feed["phase"] == 0 else float(ur[b, t])
                if feed["phase"] == 1:
                    feed["try"] = min(t + 1, 9)
                feed["phase"] ^= 1
                return self.fill_(float(lo) + (float(hi) - float(lo)) * u)

            def fake_randint(low, high=None, size=(1,), **k):
                if high is None:
                    low, high = 0, low
                u = float(ui[b]) if feed["pos"] == 0 else float(uj[b])
                feed["pos"] += 1
                return torch.full(tuple(size), min(int(math.floor(u * (high - low))) + low, high - 1), dtype=torch.int64)

            monkeypatch.setattr(torch.Tensor, "uniform_", fake_uniform)
            monkeypatch.setattr(torch, "randint", fake_randint)
            try:
                i, j, h, w = T.RandomResizedCrop.get_params(torch.zeros((3, H, W)), list(scale), list(ratio))
            finally:
                monkeypatch.undo()
            assert (int(i), int(j), int(h), int(w)) == tuple(int(v) for v in boxes[b]), (H, W, b)


@pytest.mark.parametrize("hw", [(256, 256), (240, 320), (320, 240), (300, 300), (231, 517)])
def test_resize_center_crop_restatement_matches_torchvision(hw):
    """transforms.Resize(256) + CenterCrop(224) on obs / 255 (models_r3m.py:87-89) against oracle.r3m_ref.resize_center_crop_ref. The
    reference's torchvision (0.8.2) resizes tensors with plain bilinear interpolation: newer releases are asked for antialias=False."""
    from oracle import detgen, r3m_ref
    T = tv.transforms
    H, W = hw
    x = torch.from_numpy(detgen.frames(f"tvpin_rc_{H}x{W}", (2, 3, H, W)))
    try:
        resize = T.Resize(256, antialias=False)
    except TypeError:
        resize = T.Resize(256)
    ref = T.Compose([resize, T.CenterCrop(224)])(x / 255.0) * 255.0
    got = r3m_ref.resize_center_crop_ref(x)
    assert got.shape == ref.shape == (2, 3, 224, 224)
    torch.testing.assert_close(got, ref, rtol=0, atol=2e-4)
