"""not gpu: the CPU oracle (oracle/r3m_ref.py + oracle/resnet_ref.py) against the golden vectors produced by the
REFERENCE's own code (tests/golden/make_golden.py, by-path import). This is what pins the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

from util import rel_err


def _ref_model(size, **kw):
    from oracle import detgen, r3m_ref
    m = r3m_ref.R3MRef(size=size, **kw)
    shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
    m.convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict(shapes).items()})
    return m


@pytest.mark.parametrize("size", [18, 50])
def test_oracle_encoder_matches_reference_golden(golden_dir, size):
    from oracle import detgen
    torch.set_num_threads(8)
    g = np.load(os.path.join(golden_dir, f"encoder_r{size}.npz"))
    m = _ref_model(size, langweight=0.0, tcnweight=1.0)
    assert sum(p.numel() for p in m.convnet.parameters()) == {18: 11176512, 50: 23508032}[size]
    x = torch.from_numpy(detgen.frames("frames8", (8, 3, 224, 224)))
    m.eval()
    with torch.no_grad():
        assert rel_err(m(x).numpy(), g["h_eval"])[0] < 1e-5
    m.train()
    h = m(x)
    assert rel_err(h.detach().numpy(), g["h_train"])[0] < 1e-5
    if size == 18:
        cw = torch.from_numpy(detgen.uniform("cw", tuple(h.shape), 0.5, 1.5))
        (h * cw).sum().backward()
        P = dict(m.convnet.named_parameters())
        for name, ref in zip(g["grad_names"], g["grad_norms"]):
            assert abs(float(P[str(name)].grad.double().norm()) - ref) <= 1e-3 * ref
        assert rel_err(P["conv1.weight"].grad.numpy(), g["grad_conv1.weight"])[1] < 1e-3


@pytest.mark.parametrize("l2dist", [True, False])
def test_oracle_loss_matches_reference_golden(golden_dir, l2dist):
    from oracle import detgen, r3m_ref
    sys.path.insert(0, golden_dir)
    from make_golden import make_alle
    g = np.load(os.path.join(golden_dir, f"loss_{'l2' if l2dist else 'cos'}.npz"))
    B, D = 8, 512
    m = r3m_ref.R3MRef(size=18, l2weight=1e-5, l1weight=1e-5, langweight=1.0, tcnweight=1.0, l2dist=l2dist)
    sd = {}
    for k, v in m.lang_rew.state_dict().items():
        fan_in = v.shape[1] if v.dim() == 2 else m.lang_rew.state_dict()[k.replace("bias", "weight")].shape[1]
        a = 1.0 / np.sqrt(fan_in)
        sd[k] = torch.from_numpy(detgen.uniform("lr" + k, tuple(v.shape), -a, a))
    m.lang_rew.load_state_dict(sd)
    alle = torch.from_numpy(make_alle(B, D, "alle")).requires_grad_(True)
    feats = torch.from_numpy(detgen.uniform("langfeat", (B, 768), -0.6, 0.6))
    mask = torch.ones(B)
    mask[5] = 0.0
    perms = torch.from_numpy(g["perms"])
    full, met, scores = r3m_ref.r3m_loss_ref(m, alle, tcn_perm=perms[9:15], lang_feats=feats, lang_mask=mask, lang_perm=perms[0:9])
    full.backward()
    ref = dict(zip([str(n) for n in g["metric_names"]], g["metric_values"]))
    assert set(met.keys()) == set(ref.keys())
    for k, v in ref.items():
        assert abs(met[k] - v) <= 1e-5 * max(1.0, abs(v)), (k, met[k], v)
    assert rel_err(scores.detach().numpy(), g["scores"])[0] < 1e-5
    assert rel_err(alle.grad.numpy(), g["dalle"])[0] < 1e-4
    assert rel_err(m.lang_rew.pred[8].weight.grad.numpy(), g["grad_pred.8.weight"])[0] < 1e-4
    for k, p in m.lang_rew.named_parameters():
        ref_n = float(g["gradnorm_" + k])
        assert abs(float(p.grad.double().norm()) - ref_n) <= 1e-4 * max(ref_n, 1e-12), k


def test_oracle_full_step_matches_reference_golden(golden_dir):
    from oracle import detgen, r3m_ref
    torch.set_num_threads(8)
    g = np.load(os.path.join(golden_dir, "step_r18.npz"))
    m = _ref_model(18, l2weight=1e-5, l1weight=1e-5, langweight=0.0, tcnweight=1.0)
    frames = torch.from_numpy(detgen.frames("stepframes", (2, 5, 3, 224, 224)))
    names = [str(n) for n in g["metric_names"]]
    for s in range(2):
        met = r3m_ref.train_step_ref(m, frames, tcn_perm=torch.from_numpy(g[f"perms_{s}"]))
        assert list(met.keys()) == names
        for k, v in zip(names, g[f"metric_values_{s}"]):
            assert abs(met[k] - v) <= 1e-4 * max(1.0, abs(v)), (s, k, met[k], v)
    sd = m.convnet.state_dict()
    assert rel_err(sd["bn1.running_var"].numpy(), g["post_bn1.running_var"])[0] < 1e-5
    assert np.abs(sd["bn1.weight"].numpy() - g["post_bn1.weight"]).max() < 4.2e-4


def test_oracle_matches_reference_on_the_kink_free_case(golden_dir):
    """G8 (tests/golden/encoder_r18_nokink.npz): the oracle restatement in fp32 reproduces the reference's own fp32 forward and
    gradient norms on the kink-free state, and in float64 the float64 vectors stored next to them; the case is kink-free."""
    from oracle import detgen, r3m_ref
    torch.set_num_threads(8)
    g = np.load(os.path.join(golden_dir, "encoder_r18_nokink.npz"))
    assert float(g["min_abs_z"]) > 1e-3 and 0.0 < float(g["frac_z_negative"]) < 0.01      # no kink, but the last ReLU is not trivial
    m = r3m_ref.R3MRef(size=18, langweight=0.0, tcnweight=1.0)
    shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
    sd = detgen.resnet_state_dict_no_kink(shapes, 18, tag="nk2", shift=4.0)
    m.convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    x = torch.from_numpy(detgen.frames("frames8nk", (8, 3, 224, 224)))
    m.train()
    h = m(x)
    assert rel_err(h.detach().numpy(), g["h_train"])[0] < 1e-5
    cw = torch.from_numpy(detgen.uniform("cw", tuple(h.shape), 0.5, 1.5))
    (h * cw).sum().backward()
    P = dict(m.convnet.named_parameters())
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        assert abs(float(P[str(name)].grad.double().norm()) - ref) <= 1e-3 * ref, name
    # every golden file of the family is kink-free
    for size in (34, 50):
        assert float(np.load(os.path.join(golden_dir, f"encoder_r{size}_nokink.npz"))["min_abs_z"]) > 1e-3


def test_oracle_full_step_r34_matches_reference_golden(golden_dir):
    """G5 at ResNet-34 (full-step goldens for ResNet-34 / 50 at B = 2; the ResNet-50 file is checked on the GPU)."""
    from oracle import detgen, r3m_ref
    torch.set_num_threads(8)
    g = np.load(os.path.join(golden_dir, "step_r34.npz"))
    m = _ref_model(34, l2weight=1e-5, l1weight=1e-5, langweight=0.0, tcnweight=1.0)
    frames = torch.from_numpy(detgen.frames("stepframes", (2, 5, 3, 224, 224)))
    names = [str(n) for n in g["metric_names"]]
    met = r3m_ref.train_step_ref(m, frames, tcn_perm=torch.from_numpy(g["perms_0"]))
    for k, v in zip(names, g["metric_values_0"]):
        assert abs(met[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, met[k], v)


def test_bf16_emulation_checker_is_sane():
    """oracle/bf16_emul.py (the checker of the mixed-precision encoder): it must actually round (differs from the exact
    float64 graph), stay at bf16 distance from it on a well-conditioned case (ResNet-18, eval-mode BatchNorm), keep weight
    gradients wide and activation gradients rounded."""
    from oracle import bf16_emul, resnet_ref
    torch.manual_seed(3)
    net = resnet_ref.resnet18().double()
    net.eval()
    x = torch.randn(2, 3, 224, 224, dtype=torch.float64)

    def exact(v):
        z = net.maxpool(net.relu(net.bn1(net.conv1(v))))
        return net.layer4(net.layer3(net.layer2(net.layer1(z)))).mean((2, 3))

    h_ex = exact(x)
    h_em = bf16_emul.forward_bf16(net, x)
    rel = float((h_em - h_ex).norm() / h_ex.norm())
    assert 1e-4 < rel < 2e-2, rel
    h_em.sum().backward()
    g = net.layer1[0].conv1.weight.grad
    assert g is not None and torch.isfinite(g).all()
    assert not torch.equal(g, g.to(torch.bfloat16).to(g.dtype))          # weight gradients are NOT rounded to bf16
    q = bf16_emul._QAct.apply(torch.tensor([1.0 + 2.0 ** -10], dtype=torch.float64, requires_grad=True))
    assert float(q) == 1.0                                               # 2^-10 is below bf16 resolution at 1.0


@pytest.mark.parametrize("size", [18, 34, 50])
def test_kink_table_is_consistent_with_fp64_golden(golden_dir, size):
    """tests/golden/encoder_r*_kink.npz (make_golden.encoder_kink_golden): near-zero pre-activations of the last block in the
    float64 oracle and their terms in the last BatchNorm's gradients. Cheap consistency checks here; for ResNet-34 the table
    must contain the element whose flip is the 1.344e-3 / 1.807e-4 that round 1's parity report showed for the HIP path."""
    k = np.load(os.path.join(golden_dir, f"encoder_r{size}_kink.npz"))
    g64 = np.load(os.path.join(golden_dir, f"encoder_r{size}_fp64.npz"))
    lb = "layer4.2.bn3" if size == 50 else ("layer4.1.bn2" if size == 18 else "layer4.2.bn2")
    C = g64["grad_" + lb + ".weight"].shape[0]
    n = len(k["idx"])
    assert n > 0 and all(len(k[key]) == n for key in ("channel", "z", "dgamma", "dbeta"))
    assert np.all(np.abs(k["z"]) < float(k["tau"])) and np.all(np.diff(np.abs(k["z"])) >= 0)
    assert np.all(k["channel"] == (k["idx"] // 49) % C) and np.all((k["dbeta"] > 0.5 / 49) & (k["dbeta"] < 1.5 / 49))
    if size == 34:
        ng = np.linalg.norm(g64["grad_" + lb + ".weight"].astype(np.float64))
        nb = np.linalg.norm(g64["grad_" + lb + ".bias"].astype(np.float64))
        eff = [(abs(dg) / ng, db / nb) for dg, db in zip(k["dgamma"], k["dbeta"])]
        assert any(abs(a - 1.344e-3) < 2e-6 and abs(b - 1.807e-4) < 2e-7 for a, b in eff), eff


@pytest.mark.parametrize("D", [512, 2048])
def test_oracle_language_reward_matches_reference_golden(golden_dir, D):
    """G4: the oracle's LanguageRewardRef against the reference's LanguageReward forward + backward at both head
    widths — this is what pins the D = 2048 oracle that tests/test_gpu_fullsize.py checks the HIP head against."""
    from oracle import detgen, r3m_ref
    g = np.load(os.path.join(golden_dir, f"langrew_d{D}.npz"))
    rew = r3m_ref.LanguageRewardRef(D, 1024, 768)
    full = rew.state_dict()
    sd = {}
    for k, v in full.items():
        fan_in = v.shape[1] if v.dim() == 2 else full[k.replace("bias", "weight")].shape[1]
        sd[k] = torch.from_numpy(detgen.uniform("lr" + k, tuple(v.shape), -1.0 / np.sqrt(fan_in), 1.0 / np.sqrt(fan_in)))
    rew.load_state_dict(sd)
    B = 4
    e0 = torch.from_numpy(np.maximum(detgen.uniform(f"g4e0_{D}", (B, D), -0.3, 1.0), 0)).requires_grad_(True)
    eg = torch.from_numpy(np.maximum(detgen.uniform(f"g4eg_{D}", (B, D), -0.3, 1.0), 0)).requires_grad_(True)
    le = torch.from_numpy(detgen.uniform(f"g4le_{D}", (B, 768), -0.6, 0.6))
    score = rew(e0, eg, le)
    (score * torch.from_numpy(detgen.uniform("g4cw", (B,), 0.5, 1.5))).sum().backward()
    assert rel_err(score.detach().numpy(), g["score"])[0] < 1e-5
    assert rel_err(e0.grad.numpy(), g["de0"])[0] < 1e-4 and rel_err(eg.grad.numpy(), g["deg"])[0] < 1e-4
    for k, p in rew.named_parameters():
        ref_n = float(g["gradnorm_" + k])
        assert abs(float(p.grad.double().norm()) - ref_n) <= 1e-4 * max(ref_n, 1e-12), k


def test_adam_golden_is_torch_adam(golden_dir):
    """G6: the committed Adam trajectory (torch.optim.Adam, lr 1e-4, defaults — models_r3m.py:76) reproduces with the formula of
    SURVEY.md §8(a) A11 evaluated in float64; the GPU test gates the fused HIP Adam against the same file."""
    from oracle import detgen
    g = np.load(os.path.join(golden_dir, "adam.npz"))
    p = detgen.uniform("g6p", (4096,), -1.0, 1.0).astype(np.float64)
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for i in range(3):
        gr = (detgen.uniform(f"g6g{i}", (4096,), -1.0, 1.0) * np.float32(10.0 ** (i - 1))).astype(np.float64)
        m = 0.9 * m + 0.1 * gr
        v = 0.999 * v + 0.001 * gr * gr
        t = i + 1
        p = p - 1e-4 / (1 - 0.9 ** t) * m / (np.sqrt(v) / np.sqrt(1 - 0.999 ** t) + 1e-8)
        assert np.abs(p - g[f"p_{i}"]).max() < 2e-7
    assert rel_err(m, g["exp_avg"])[0] < 1e-6 and rel_err(v, g["exp_avg_sq"])[0] < 1e-6
