"""-m gpu: the HEADLINE size (BASELINE configs[1]/[2]: ResNet-50, 256 clips = 1280 frames per GPU) through properties that do
not need an oracle at that size — the oracle-pinned small-batch path is the reference point:

  * eval mode: a frame's embedding does not depend on what else is in the batch, so rows of the 1280-frame result must equal
    the same frames pushed through an 8-frame plan (different tile counts, split-K factors, arena offsets, block orders);
  * train mode: BatchNorm batch statistics are permutation invariant -> permuting the frames permutes the embeddings;
  * backward is linear in the output gradient: backward(dh1) followed by an ACCUMULATING backward(dh2) equals backward(dh1 + dh2);
  * every parameter gradient is finite and non-trivial.
Tolerances are fp32 summation-order level for fp32 and one bf16 rounding step for the bf16 plan."""
import numpy as np
import pytest
import torch

from util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F_FULL = 1280


def _model(precision):
    from r3m_amd import R3M
    torch.manual_seed(5)
    m = R3M("cuda", 1e-4, 1024, size=50, langweight=0.0, tcnweight=1.0, precision=precision).to(DEV)
    # calibrated running statistics so that eval mode is well scaled: one train-mode pass over a few frames
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randint(0, 256, (64, 3, 224, 224), generator=g, device=DEV, dtype=torch.int32).float()
    m.train()
    with torch.no_grad():
        for _ in range(3):
            m(x)
    return m


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_headline_size_properties(hip, precision):
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs the 288 GB of an MI355X")
    m = _model(precision)
    tol = 2e-5 if precision == "fp32" else 2.0 ** -7
    g = torch.Generator(device=DEV).manual_seed(10)
    x = torch.randint(0, 256, (F_FULL, 3, 224, 224), generator=g, device=DEV, dtype=torch.int32).float()

    # ---- eval: batch-composition invariance against the small plan ----
    m.eval()
    with torch.no_grad():
        h_full = m(x).clone()
        idx = torch.tensor([0, 1, 7, 128, 255, 640, 1000, 1279], device=DEV)
        h_small = m(x[idx]).clone()
    assert torch.isfinite(h_full).all() and float(h_full.abs().max()) > 0
    e_max, e_l2 = rel_err(h_full[idx].cpu().numpy(), h_small.cpu().numpy())
    assert e_max <= tol, f"{precision}: eval rows differ between the 1280-frame and the 8-frame plan: {e_max}"

    # ---- train: permutation equivariance through the batch statistics ----
    m.train()
    perm = torch.randperm(F_FULL, generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        h_a = m(x).clone()
        h_b = m(x[perm]).clone()
    e_max, e_l2 = rel_err(h_b.cpu().numpy(), h_a[perm].cpu().numpy())
    # fp32: summation-order level. bf16: a permutation changes the fp32 statistics partials in the last bits, which flips a few
    # bf16 roundings of the normalised activations; on i.i.d.-noise frames at initialisation those flips are amplified layer by
    # layer (tests/test_gpu_bf16.py discusses the conditioning) — measured 7e-2 max / 2e-2 l2, gated loosely.
    print(f"{precision}: train-mode permutation equivariance max-rel {e_max:.3e} l2-rel {e_l2:.3e}")
    if precision == "fp32":
        assert e_max <= 1e-4, f"fp32: train-mode permutation equivariance {e_max}"
    else:
        assert e_l2 <= 6e-2 and e_max <= 0.25, f"bf16: train-mode permutation equivariance max {e_max} l2 {e_l2}"

    # ---- backward: linearity in dh, through the accumulate path ----
    h = m(x)
    gd = torch.Generator(device=DEV).manual_seed(11)
    dh1 = torch.rand(h.shape, generator=gd, device=DEV) - 0.5
    dh2 = torch.rand(h.shape, generator=gd, device=DEV) - 0.5
    m.encoder_opt.zero_grad()
    h.backward(dh1, retain_graph=True)
    h.backward(dh2, retain_graph=True)                      # accumulates into the flat gradient buffer
    g_acc = m.convnet.flat_grads().clone()
    m.encoder_opt.zero_grad()
    h.backward(dh1 + dh2)
    g_sum = m.convnet.flat_grads().clone()
    assert torch.isfinite(g_sum).all()
    P = dict(m.convnet.named_parameters())
    assert all(float(p.grad.abs().max()) > 0 for p in P.values())
    num = float((g_acc - g_sum).double().norm())
    den = float(g_sum.double().norm())
    # fp32: summation-order noise only (measured 7.6e-6); bf16: the activation gradients are rounded separately in the two passes (2.2e-2)
    print(f"{precision}: backward linearity |g(dh1)+g(dh2) - g(dh1+dh2)| / |g| = {num / den:.3e}")
    assert num / den <= (1e-4 if precision == "fp32" else 1e-1), f"{precision}: backward linearity {num / den}"
    del m, x, h
    torch.cuda.empty_cache()
