"""Generates tests/golden/*.npz by running the REFERENCE's own code (imported by path, oracle/by_path.py) on CPU fp32.
Run in the authoring container only (needs /root/reference):   python tests/golden/make_golden.py
Inputs and weights come from oracle/detgen.py (hash generator) so they are NOT stored — only the expected outputs are.

  G1/G2  encoder_r{18,34,50}.npz : reference R3M.forward on 8 frames, train and eval mode; running stats; gradients of
                                   L = sum(h * cw) (full conv1/bn1 grads + L2 norm of every parameter gradient)
  G3     loss_{l2,cos}.npz       : reference Trainer.update on given embeddings (fake encoder), TCN + LP + language
                                   InfoNCE with the reference LanguageReward: metrics, d full_loss/d alle, scores, head grads
  G5     step_r{18,34,50}.npz    : two full reference Trainer.update steps (R3M + Adam), B=2 clips
  G9     step_r{18,34,50}_lr1e-7.npz : G5 with lr = 1e-7 (no Adam sign chaos): both steps' metrics, running statistics, Adam moments
  G8     encoder_r{18,34,50}_nokink.npz : the G1/G2 gradient case on the KINK-FREE state (detgen.resnet_state_dict_no_kink, tag
                                   "nk2", shift 4.0; tools/experiments/find_nokink.py): no float64 pre-activation of the last
                                   block lies within 1.3e-3 of zero, so every fp32 forward makes the float64 ReLU decisions
                                   there and the gradient gate (3x the reference's own fp32 error) applies with NO flip
                                   accounting. Holds the reference's fp32 results AND the float64 oracle's.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import by_path, detgen  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def set_state(model_convnet, tag="w"):
    shapes = [(k, tuple(v.shape)) for k, v in model_convnet.state_dict().items()]
    sd = detgen.resnet_state_dict(shapes, tag)
    model_convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})


def last_bn_name(size):
    return "layer4.2.bn3" if size == 50 else ("layer4.1.bn2" if size == 18 else "layer4.2.bn2")


def encoder_golden(size, F=8):
    r3m, _, _ = by_path.load_reference()
    m = r3m.R3M("cpu", 1e-4, 1024, size=size, langweight=0.0, tcnweight=1.0)
    set_state(m.convnet)
    x = torch.from_numpy(detgen.frames(f"frames{F}", (F, 3, 224, 224)))
    out = {}
    m.eval()
    with torch.no_grad():
        out["h_eval"] = m(x).numpy()
    m.train()
    h = m(x)
    out["h_train"] = h.detach().numpy()
    sd = m.convnet.state_dict()
    lb = last_bn_name(size)
    for k in ("bn1.running_mean", "bn1.running_var", lb + ".running_mean", lb + ".running_var"):
        out["post_" + k] = sd[k].numpy().copy()
    cw = torch.from_numpy(detgen.uniform("cw", tuple(h.shape), 0.5, 1.5))
    (h * cw).sum().backward()
    names, norms = [], []
    for k, p in m.convnet.named_parameters():
        names.append(k)
        norms.append(float(p.grad.double().norm()))
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, dtype=np.float64)
    P = dict(m.convnet.named_parameters())
    for k in ("conv1.weight", "bn1.weight", "bn1.bias", lb + ".weight", lb + ".bias", "layer1.0.conv1.weight",
              "layer2.0.downsample.0.weight"):
        out["grad_" + k] = P[k].grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"encoder_r{size}.npz"), **out)
    print("encoder", size, out["h_train"].shape, float(np.abs(out["h_train"]).max()))


def encoder_fp64_golden(size, F=8):
    """Noise-floor reference: the SAME graph evaluated in float64 with the oracle restatement (oracle/r3m_ref.py). Used to
    express gradient tolerances relative to the reference's own fp32 round-off (its fp32 gradients sit ~5e-3 l2-rel away
    from fp64 at conv1 for ResNet-18, more for deeper nets)."""
    from oracle import r3m_ref
    g32 = np.load(os.path.join(OUT, f"encoder_r{size}.npz"))
    m = r3m_ref.R3MRef(size=size, langweight=0.0, tcnweight=1.0)
    set_state(m.convnet)
    m = m.double()
    x = torch.from_numpy(detgen.frames(f"frames{F}", (F, 3, 224, 224))).double()
    m.train()
    h = m.convnet(m.normlayer(x / 255.0))
    cw = torch.from_numpy(detgen.uniform("cw", tuple(h.shape), 0.5, 1.5)).double()
    (h * cw).sum().backward()
    P = dict(m.convnet.named_parameters())
    lb = last_bn_name(size)
    out = {"h_train": h.detach().numpy(), "grad_names": g32["grad_names"],
           "grad_norms": np.array([float(P[str(n)].grad.norm()) for n in g32["grad_names"]])}
    for k in ("conv1.weight", "bn1.weight", "bn1.bias", lb + ".weight", lb + ".bias", "layer1.0.conv1.weight",
              "layer2.0.downsample.0.weight"):
        out["grad_" + k] = P[k].grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f"encoder_r{size}_fp64.npz"), **out)
    e = np.linalg.norm(g32["grad_conv1.weight"].astype(np.float64) - P["conv1.weight"].grad.numpy()) / np.linalg.norm(P["conv1.weight"].grad.numpy())
    print("fp64", size, "reference-fp32 conv1.weight grad l2-rel vs fp64:", e)


GRAD_KEYS = lambda lb: ("conv1.weight", "bn1.weight", "bn1.bias", lb + ".weight", lb + ".bias", "layer1.0.conv1.weight",  # noqa: E731
                        "layer2.0.downsample.0.weight")


def encoder_nokink_golden(size, F=8, draw=0):
    """G8: reference fp32 (its own R3M.forward + autograd) and float64 oracle on a kink-free state, in one file. `draw` picks one of
    the three independent (weights, frames) cases of detgen.NOKINK_STATES[size]: the gradient gate of tests/test_gpu_encoder.py is
    statistical (median of the three HIP / reference error ratios, plus a bound per draw)."""
    from oracle import r3m_ref
    r3m, _, _ = by_path.load_reference()
    lb = last_bn_name(size)
    nk_tag, nk_shift, nk_frames = detgen.NOKINK_STATES[size][draw]

    def state(convnet):
        shapes = [(k, tuple(v.shape)) for k, v in convnet.state_dict().items()]
        sd = detgen.resnet_state_dict_no_kink(shapes, size, tag=nk_tag, shift=nk_shift)
        convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})

    x = torch.from_numpy(detgen.frames(nk_frames, (F, 3, 224, 224)))
    out = {}
    # the reference, fp32
    m = r3m.R3M("cpu", 1e-4, 1024, size=size, langweight=0.0, tcnweight=1.0)
    state(m.convnet)
    m.train()
    h = m(x)
    out["h_train"] = h.detach().numpy()
    cw = torch.from_numpy(detgen.uniform("cw", tuple(h.shape), 0.5, 1.5))
    (h * cw).sum().backward()
    P = dict(m.convnet.named_parameters())
    names = list(P.keys())
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array([float(P[k].grad.double().norm()) for k in names], dtype=np.float64)
    for k in GRAD_KEYS(lb):
        out["grad_" + k] = P[k].grad.numpy().copy()
    # the oracle, float64 (+ the last block's pre-activations: how far from a kink the case is)
    o = r3m_ref.R3MRef(size=size, langweight=0.0, tcnweight=1.0)
    state(o.convnet)
    o = o.double()
    o.train()
    net = o.convnet
    blk = net.layer4[-1]
    last = blk.bn3 if size == 50 else blk.bn2
    keep = {}
    hooks = [last.register_forward_hook(lambda mod, i, o_: keep.__setitem__("bn_out", o_.detach())),
             blk.register_forward_pre_hook(lambda mod, i: keep.__setitem__("idn", i[0].detach()))]
    h64 = net(o.normlayer(x.double() / 255.0))
    for hk in hooks:
        hk.remove()
    z = (keep["bn_out"] + keep["idn"]).flatten()
    out["min_abs_z"] = np.array(float(z.abs().min()))
    out["frac_z_negative"] = np.array(float((z < 0).double().mean()))
    assert float(z.abs().min()) > 1e-3, "not kink-free: rerun tools/experiments/find_nokink.py"
    (h64 * cw.double()).sum().backward()
    P64 = dict(net.named_parameters())
    out["h_train_fp64"] = h64.detach().numpy()
    out["grad_norms_fp64"] = np.array([float(P64[k].grad.norm()) for k in names], dtype=np.float64)
    for k in GRAD_KEYS(lb):
        out["grad64_" + k] = P64[k].grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f"encoder_r{size}_nokink{detgen.NOKINK_SUFFIX[draw]}.npz"), **out)
    e = np.linalg.norm(out["grad_conv1.weight"].astype(np.float64) - P64["conv1.weight"].grad.numpy()) / np.linalg.norm(P64["conv1.weight"].grad.numpy())
    print("nokink", size, "min|z|", float(out["min_abs_z"]), "frac z<0", float(out["frac_z_negative"]), "reference-fp32 conv1 grad vs fp64:", e)


def encoder_kink_golden(size, F=8, tau=2e-4):
    """ReLU-kink table of the LAST block (float64 oracle): the elements of its pre-ReLU sum z = bn(y) + identity with
    |z| < tau, and what each contributes to the last BatchNorm's d(gamma) / d(beta) when its ReLU is on. Any fp32
    implementation may decide such an element the other way (the forward's own round-off is ~1e-5 of the activation scale);
    ONE flipped element moves d(gamma) of ResNet-34's last BatchNorm by 1.3e-3 l2-rel (tools/experiments/debug_r34_lastbn.py:
    exactly the error VERDICT r1 weak #3 saw). The GPU test uses this table to gate the last-BN gradients at 1e-4 UP TO
    those decisions instead of hiding them under a 3e-3 floor."""
    from oracle import r3m_ref
    m = r3m_ref.R3MRef(size=size, langweight=0.0, tcnweight=1.0)
    set_state(m.convnet)
    m = m.double()
    x = torch.from_numpy(detgen.frames(f"frames{F}", (F, 3, 224, 224))).double()
    m.train()
    net = m.convnet
    blk = net.layer4[-1]
    last = blk.bn3 if size == 50 else blk.bn2
    keep = {}
    hooks = [last.register_forward_hook(lambda mod, i, o: keep.__setitem__("bn_out", o.detach())),
             last.register_forward_pre_hook(lambda mod, i: keep.__setitem__("bn_in", i[0].detach())),
             blk.register_forward_pre_hook(lambda mod, i: keep.__setitem__("idn", i[0].detach()))]
    h = net(m.normlayer(x / 255.0))
    for hk in hooks:
        hk.remove()
    cw = torch.from_numpy(detgen.uniform("cw", tuple(h.shape), 0.5, 1.5)).double()
    z = keep["bn_out"] + keep["idn"]
    y = keep["bn_in"]
    yhat = (y - y.mean((0, 2, 3), keepdim=True)) / torch.sqrt(y.var((0, 2, 3), unbiased=False, keepdim=True) + 1e-5)
    dz = (cw / (z.shape[2] * z.shape[3])).view(F, -1, 1, 1).expand_as(z)
    idx = torch.nonzero(z.abs().flatten() < tau).flatten()
    idx = idx[torch.argsort(z.flatten()[idx].abs())]
    chan = (idx // (z.shape[2] * z.shape[3])) % z.shape[1]
    out = {"tau": np.array(tau), "idx": idx.numpy(), "channel": chan.numpy(), "z": z.flatten()[idx].numpy(),
           "dgamma": (dz * yhat).flatten()[idx].numpy(), "dbeta": dz.flatten()[idx].numpy()}
    np.savez_compressed(os.path.join(OUT, f"encoder_r{size}_kink.npz"), **out)
    print("kink", size, len(idx), "elements with |z| <", tau, "closest", float(out["z"][0]) if len(idx) else None)


class _FakeCore(torch.nn.Module):
    """Stands where `model.module` does in the reference Trainer: loss weights, sim, get_reward with fixed text features."""

    def __init__(self, ref_r3m_mod, ref_lang_mod, D, l2dist, langweight, feats):
        super().__init__()
        self.l2weight, self.l1weight, self.tcnweight, self.langweight = 1e-5, 1e-5, 1.0, langweight
        self.num_negatives = 3
        self.l2dist = l2dist
        self.cs = torch.nn.CosineSimilarity(1)
        self.lang_rew = ref_lang_mod.LanguageReward(None, D, 1024, 768)
        self.feats = feats
        self._sim = ref_r3m_mod.R3M.sim

    def sim(self, a, b):
        return self._sim(self, a, b)     # the reference's own R3M.sim body

    def get_reward(self, e0, es, sentences):
        return self.lang_rew(e0, es, self.feats)

    class _Opt:
        def zero_grad(self): pass
        def step(self): pass
    encoder_opt = _Opt()


class _FakeModel(torch.nn.Module):
    def __init__(self, core, alles):
        super().__init__()
        self.module = core
        self.alles = alles

    def forward(self, x):
        return self.alles


def make_alle(B, D, tag):
    u = detgen.uniform(tag, (B, 5, D), 0.0, 1.0)
    a = np.maximum(u - 0.3, 0.0) * 2.0      # ~30 % exact zeros, like avg-pooled ReLU features
    a[3, 4] = a[3, 3]                        # es2 == es1 for one clip -> s12 == 0 exactly (data_loaders.py:77-79 allows it)
    return a.astype(np.float32)


def loss_golden(l2dist, B=8, D=512):
    r3m, lang, trainer = by_path.load_reference()
    feats = torch.from_numpy(detgen.uniform("langfeat", (B, 768), -0.6, 0.6))
    core = _FakeCore(r3m, lang, D, l2dist, 1.0, feats)
    shapes = [(k, tuple(v.shape)) for k, v in core.lang_rew.state_dict().items()]
    sd = {}
    for k, shp in shapes:
        fan_in = shp[1] if len(shp) == 2 else shapes[[s[0] for s in shapes].index(k.replace("bias", "weight"))][1][1]
        a = 1.0 / np.sqrt(fan_in)
        sd[k] = torch.from_numpy(detgen.uniform("lr" + k, shp, -a, a))
    core.lang_rew.load_state_dict(sd)
    alles = torch.from_numpy(make_alle(B, D, "alle").reshape(B * 5, D)).requires_grad_(True)
    model = _FakeModel(core, alles)
    b_lang = ["open the drawer"] * B
    b_lang[5] = ""                           # masked clip (trainer.py:107-109)
    seed = 1234
    torch.manual_seed(seed)
    perms = torch.stack([torch.randperm(B) for _ in range(15)])   # the draws Trainer.update is about to make
    torch.manual_seed(seed)
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self                # trainer.py:108 hard-codes .cuda()
    try:
        metrics, _ = trainer.Trainer(1).update(model, (torch.zeros(B, 5, 3, 224, 224), b_lang), 0)
    finally:
        torch.Tensor.cuda = orig_cuda
    out = {"perms": perms.numpy(), "dalle": alles.grad.numpy().reshape(B, 5, D).copy(),
           "metric_names": np.array(list(metrics.keys())), "metric_values": np.array(list(metrics.values()), dtype=np.float64)}
    for k, p in core.lang_rew.named_parameters():
        out["gradnorm_" + k] = np.array(float(p.grad.double().norm()))
    out["grad_pred.8.weight"] = core.lang_rew.pred[8].weight.grad.numpy().copy()
    out["grad_pred.0.bias"] = core.lang_rew.pred[0].bias.grad.numpy().copy()
    # scores of the 15 reward evaluations in call order, recomputed with the same perms (no grad)
    with torch.no_grad():
        alle = alles.detach().reshape(B, 5, D)
        e0, eg, es0, es1, es2 = [alle[:, i] for i in range(5)]
        G = lambda a, b: core.lang_rew(a, b, feats)[0]
        sc = [G(e0, eg), G(e0, es1), G(e0, es2), G(e0, e0), G(e0, es0), G(e0, es1)]
        for k in range(3):
            for j, other in enumerate((eg, es1, es2)):
                p = perms[3 * k + j]
                sc.append(G(e0[p], other[p]))
        out["scores"] = torch.stack(sc).numpy()
    np.savez_compressed(os.path.join(OUT, f"loss_{'l2' if l2dist else 'cos'}.npz"), **out)
    print("loss", "l2" if l2dist else "cos", dict(zip(metrics.keys(), [round(v, 6) for v in metrics.values()])))


def step_golden(size=18, B=2, nsteps=2):
    r3m, _, trainer = by_path.load_reference()
    m = r3m.R3M("cpu", 1e-4, 1024, size=size, l2weight=1e-5, l1weight=1e-5, langweight=0.0, tcnweight=1.0)
    set_state(m.convnet)

    class Wrap(torch.nn.Module):
        def __init__(self, mod):
            super().__init__()
            self.module = mod

        def forward(self, x):
            return self.module(x)

    model = Wrap(m)
    frames = torch.from_numpy(detgen.frames("stepframes", (B, 5, 3, 224, 224)))
    out = {}
    seed = 77
    torch.manual_seed(seed)
    perms = [torch.stack([torch.randperm(B) for _ in range(6)]) for _ in range(nsteps)]
    torch.manual_seed(seed)
    T = trainer.Trainer(1)
    lb = last_bn_name(size)
    last_conv = lb.replace("bn", "conv")
    for s in range(nsteps):
        metrics, _ = T.update(model, (frames, [""] * B), s)
        out[f"metric_values_{s}"] = np.array(list(metrics.values()), dtype=np.float64)
        out["metric_names"] = np.array(list(metrics.keys()))
        out[f"perms_{s}"] = perms[s].numpy()
        if s == 0:     # weights after the FIRST Adam step (every weight moved by ~lr with the sign of its gradient)
            sd1 = m.convnet.state_dict()
            for k in ("conv1.weight", "bn1.weight", "bn1.bias", lb + ".weight"):
                out["post1_" + k] = sd1[k].numpy().copy()
    sd = m.convnet.state_dict()
    for k in ("conv1.weight", "bn1.weight", "bn1.bias", "bn1.running_mean", "bn1.running_var", lb + ".weight",
              last_conv + ".weight", "layer1.0.conv1.weight"):
        out["post_" + k] = sd[k].numpy().copy() if sd[k].numel() < 50000 else sd[k].numpy().reshape(-1)[:50000].copy()
    out["nbt"] = sd["bn1.num_batches_tracked"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"step_r{size}.npz"), **out)
    print("step", [out[f"metric_values_{s}"] for s in range(nsteps)])


def step_small_lr_golden(size, B=2, nsteps=2, lr=1e-7):
    """G9: the G5 case with lr = 1e-7. G5's lr = 1e-4 makes step 2 chaotic by construction (Adam's first step moves every weight by
    +-lr with the SIGN of its gradient, round-off-level gradients flip sign between any two fp32 implementations); with 1e-7 the
    weights after step 1 agree to ~2e-7 whatever the signs, so BOTH steps' metrics, the running statistics after two train-mode
    forwards, `num_batches_tracked` and Adam's moments (exp_avg = 0.1 g, exp_avg_sq = 0.001 g^2 after step 1: a direct image of the
    conv1 gradient) are comparable at fp32 tolerances. Made by the reference's own Trainer.update / torch.optim.Adam
    (/root/reference/r3m/trainer.py:155-158, models_r3m.py:76)."""
    r3m, _, trainer = by_path.load_reference()
    m = r3m.R3M("cpu", lr, 1024, size=size, l2weight=1e-5, l1weight=1e-5, langweight=0.0, tcnweight=1.0)
    set_state(m.convnet)

    class Wrap(torch.nn.Module):
        def __init__(self, mod):
            super().__init__()
            self.module = mod

        def forward(self, x):
            return self.module(x)

    model = Wrap(m)
    frames = torch.from_numpy(detgen.frames("stepframes", (B, 5, 3, 224, 224)))
    out = {"lr": np.array(lr)}
    seed = 77
    torch.manual_seed(seed)
    T = trainer.Trainer(1)
    lb = last_bn_name(size)
    P = dict(m.convnet.named_parameters())
    for s in range(nsteps):
        metrics, _ = T.update(model, (frames, [""] * B), s)
        out[f"metric_values_{s}"] = np.array(list(metrics.values()), dtype=np.float64)
        out["metric_names"] = np.array(list(metrics.keys()))
        st = m.encoder_opt.state[P["conv1.weight"]]
        out[f"exp_avg_conv1_{s}"] = st["exp_avg"].numpy().copy()
        out[f"exp_avg_sq_conv1_{s}"] = st["exp_avg_sq"].numpy().copy()
        stb = m.encoder_opt.state[P[lb + ".weight"]]
        out[f"exp_avg_lastbn_{s}"] = stb["exp_avg"].numpy().copy()
        out[f"exp_avg_sq_lastbn_{s}"] = stb["exp_avg_sq"].numpy().copy()
    sd = m.convnet.state_dict()
    for k in ("bn1.running_mean", "bn1.running_var", lb + ".running_mean", lb + ".running_var", "bn1.weight", lb + ".weight"):
        out["post_" + k] = sd[k].numpy().copy()
    out["post_conv1.weight"] = sd["conv1.weight"].numpy().copy()
    out["nbt"] = sd["bn1.num_batches_tracked"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"step_r{size}_lr1e-7.npz"), **out)
    print("step lr=1e-7", size, [out[f"metric_values_{s}"] for s in range(nsteps)])


def lang_state(module):
    sd, full = {}, module.state_dict()
    for k, v in full.items():
        fan_in = v.shape[1] if v.dim() == 2 else full[k.replace("bias", "weight")].shape[1]
        a = 1.0 / np.sqrt(fan_in)
        sd[k] = torch.from_numpy(detgen.uniform("lr" + k, tuple(v.shape), -a, a))
    return sd


def language_reward_golden(D, B=4):
    """G4: the reference's LanguageReward (models_language.py:37-55) forward + backward at D = 512 and D = 2048."""
    _, lang, _ = by_path.load_reference()
    rew = lang.LanguageReward(None, D, 1024, 768)
    rew.load_state_dict(lang_state(rew))
    e0 = torch.from_numpy(np.maximum(detgen.uniform(f"g4e0_{D}", (B, D), -0.3, 1.0), 0)).requires_grad_(True)
    eg = torch.from_numpy(np.maximum(detgen.uniform(f"g4eg_{D}", (B, D), -0.3, 1.0), 0)).requires_grad_(True)
    le = torch.from_numpy(detgen.uniform(f"g4le_{D}", (B, 768), -0.6, 0.6))
    score, info = rew(e0, eg, le)
    cw = torch.from_numpy(detgen.uniform("g4cw", (B,), 0.5, 1.5))
    (score * cw).sum().backward()
    out = {"score": score.detach().numpy(), "de0": e0.grad.numpy(), "deg": eg.grad.numpy(),
           "grad_pred.8.weight": rew.pred[8].weight.grad.numpy().copy(), "grad_pred.0.bias": rew.pred[0].bias.grad.numpy().copy()}
    for k, p_ in rew.named_parameters():
        out["gradnorm_" + k] = np.array(float(p_.grad.double().norm()))
    np.savez_compressed(os.path.join(OUT, f"langrew_d{D}.npz"), **out)
    print("langrew", D, out["score"])


def adam_golden(nsteps=3):
    """G6: torch.optim.Adam(lr=1e-4) — the optimizer the reference builds (models_r3m.py:76) — three steps on a small tensor."""
    n = 4096
    p = torch.from_numpy(detgen.uniform("g6p", (n,), -1.0, 1.0)).requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-4)
    out = {}
    for i in range(nsteps):
        p.grad = torch.from_numpy(detgen.uniform(f"g6g{i}", (n,), -1.0, 1.0) * np.float32(10.0 ** (i - 1)))
        opt.step()
        out[f"p_{i}"] = p.detach().numpy().copy()
    st = opt.state[p]
    out["exp_avg"], out["exp_avg_sq"] = st["exp_avg"].numpy().copy(), st["exp_avg_sq"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "adam.npz"), **out)
    print("adam", out["p_2"][:3])


def lang_encoder_golden():
    """G7: the reference's OWN LangEncoder.forward (models_language.py:23-35: tokenise, transformer, mean(1) over all positions)
    with `AutoTokenizer/AutoModel.from_pretrained` answering with the tiny stand-ins of oracle/tiny_text.py (the pretrained
    files are absent from the image): the whole batch, and a batch of two equally short sentences (no padding -> different features for
    the SAME sentences: the padding quirk)."""
    import transformers
    from oracle import tiny_text
    _, lang, _ = by_path.load_reference()
    saved = transformers.AutoTokenizer.from_pretrained, transformers.AutoModel.from_pretrained
    transformers.AutoTokenizer.from_pretrained = staticmethod(lambda *a, **k: tiny_text.WhitespaceTokenizer())
    transformers.AutoModel.from_pretrained = staticmethod(lambda *a, **k: tiny_text.tiny_distilbert())
    try:
        enc = lang.LangEncoder("cpu")
    finally:
        transformers.AutoTokenizer.from_pretrained, transformers.AutoModel.from_pretrained = saved
    enc.eval()
    out = {"feats_all": enc(tiny_text.SENTENCES).numpy(), "feats_short": enc([tiny_text.SENTENCES[0], tiny_text.SENTENCES[3]]).numpy(),
           "feats_array_input": enc(np.array(tiny_text.SENTENCES[3:6])).numpy()}
    np.savez_compressed(os.path.join(OUT, "lang_encoder_tiny.npz"), **out)
    print("lang_encoder", out["feats_all"].shape, float(np.abs(out["feats_all"][0] - out["feats_short"][0]).max()))


if __name__ == "__main__":
    assert by_path.available(), "/root/reference is required to (re)generate goldens"
    for size in (18, 34, 50):
        encoder_golden(size)
        encoder_fp64_golden(size)
        encoder_kink_golden(size)
        for draw in range(3):
            encoder_nokink_golden(size, draw=draw)
    loss_golden(True)
    loss_golden(False)
    for size in (18, 34, 50):
        step_golden(size)
        step_small_lr_golden(size)
    language_reward_golden(512)
    language_reward_golden(2048)
    adam_golden()
    lang_encoder_golden()
