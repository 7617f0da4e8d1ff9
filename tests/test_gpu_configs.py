"""-m gpu: every BASELINE.json config exercised AS a config, not only as separately tested pieces.

  configs[0]  load_r3m('resnet18') forward on 8x3x224x224        -> tests/test_gpu_encoder.py (golden G1) + tests/test_gpu_train.py
  configs[1]  ResNet-50 fp32, 256 clips, TCN                     -> tests/test_gpu_fullsize.py (headline size), bench.py
  configs[2]  ResNet-50 + langweight=1 + L1=1e-5, bf16           -> test_config2_* below (composition vs the emulated format + fp32 loss oracle)
  configs[3]  ResNet-50 full loss, DDP over RCCL                 -> tests/test_gpu_ddp.py (1-rank RCCL forced sync, torchrun bench) + test_config3_*
  configs[4]  ResNet-34 bf16, 512 clips, doaug=rctraj on the GPU -> test_config4_* below (2560 frames, crop inside the step)
"""
import numpy as np
import pytest
import torch

from util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lang_state(module):
    from oracle import detgen
    sd = {}
    full = module.state_dict()
    for k, v in full.items():
        fan_in = v.shape[1] if v.dim() == 2 else full[k.replace("bias", "weight")].shape[1]
        a = 1.0 / np.sqrt(fan_in)
        sd[k] = torch.from_numpy(detgen.uniform("lr" + k, tuple(v.shape), -a, a))
    return sd


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float(a @ b / (a.norm() * b.norm()).clamp_min(1e-300))


@pytest.mark.parametrize("size,B", [(18, 4), (50, 2)])
def test_config2_bf16_language_step_vs_emulated_oracle(hip, size, B):
    """BASELINE configs[2] as ONE Trainer.update: precision="bf16", langweight=1 (frozen text features), L1 = L2 = 1e-5, TCN,
    Adam. Checker = the composition the config implies: oracle/bf16_emul.py for the encoder (float64 with a bf16 rounding at
    every tensor the engine stores in bf16, BatchNorm on batch statistics) feeding the fp32-semantics loss oracle
    (oracle/r3m_ref.r3m_loss_ref: LP + TCN + language InfoNCE through the reward head). Well-conditioned inputs/state as in
    test_encoder_bf16_well_conditioned_absolute, so the gates are absolute."""
    from oracle import bf16_emul, detgen, r3m_ref
    from r3m_amd import R3M
    from r3m_amd.parallel import SingleDevice
    from r3m_amd.trainer import Trainer
    m = R3M("cuda", 1e-4, 1024, size=size, l2weight=1e-5, l1weight=1e-5, langweight=1.0, tcnweight=1.0, precision="bf16")
    shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict_small_residual(shapes, size, 0.1).items()}
    m.convnet.load_state_dict(sd)
    lsd = _lang_state(m.lang_rew)
    m.lang_rew.load_state_dict(lsd)
    model = SingleDevice(m).to(DEV)
    frames = torch.from_numpy(detgen.smooth_frames("cfg2", (B * 5, 3, 224, 224), 7)).reshape(B, 5, 3, 224, 224)
    feats = torch.from_numpy(detgen.uniform("langfeat", (B, 768), -0.6, 0.6))
    mask = torch.ones(B)
    mask[B - 1] = 0.0
    torch.manual_seed(5)
    lang_perm = torch.stack([torch.randperm(B) for _ in range(9)])
    tcn_perm = torch.stack([torch.randperm(B) for _ in range(6)])

    # ---- checker: emulated-bf16 encoder (train mode) -> loss oracle, float64 ----
    ref = r3m_ref.R3MRef(size=size, l2weight=1e-5, l1weight=1e-5, langweight=1.0, tcnweight=1.0)
    ref.convnet.load_state_dict(sd)
    ref.lang_rew.load_state_dict(lsd)
    ref = ref.double()
    ref.train()
    xn = ref.normlayer(frames.reshape(B * 5, 3, 224, 224).double() / 255.0)
    alles = bf16_emul.forward_bf16(ref.convnet, xn)
    full_ref, met_ref, scores_ref = r3m_ref.r3m_loss_ref(ref, alles.reshape(B, 5, -1), tcn_perm=tcn_perm, lang_feats=feats.double(),
                                                         lang_mask=mask.double(), lang_perm=lang_perm)
    ref.zero_grad()
    full_ref.backward()
    g_ref = {k: p.grad.clone() for k, p in ref.convnet.named_parameters()}
    gl_ref = {k: p.grad.clone() for k, p in ref.lang_rew.named_parameters()}

    # ---- the product path: one step ----
    before = m.convnet.flat_params().clone()
    torch.manual_seed(5)
    metrics, _ = Trainer(1).update(model, (frames.to(DEV), (feats.to(DEV), mask)), 0)
    assert set(metrics) == set(met_ref)
    line = []
    for k, v in met_ref.items():
        line.append(f"{k} {metrics[k]:.5f}/{v:.5f}")
        if k.startswith("rewacc") or k == "aligned":
            assert abs(metrics[k] - v) <= 1.0 / B + 1e-6, (k, metrics[k], v)        # a ranking count: at most one clip may differ
        else:
            assert abs(metrics[k] - v) <= 2e-2 * max(1.0, abs(v)), (k, metrics[k], v)
    print(f"configs[2] r{size} B={B}: hip/checker " + ", ".join(line))
    g_hip = {k: p.grad.detach().cpu() for k, p in m.convnet.named_parameters()}
    assert all(torch.isfinite(v).all() for v in g_hip.values())
    tot = sum(float(v.double().pow(2).sum()) for v in g_ref.values())
    dot = sum(float((g_ref[k].double() * g_hip[k].double()).sum()) for k in g_ref)
    nb = sum(float(g_hip[k].double().pow(2).sum()) for k in g_ref)
    cos_all = dot / (tot * nb) ** 0.5
    worst = min(_cos(g_ref[k], g_hip[k]) for k in g_ref if g_ref[k].dim() == 4 and float(g_ref[k].double().pow(2).sum()) > 1e-3 * tot)
    # (pred.8.bias is ONE number, the sum of d loss/d score over all 15 B scores — ~0 by the InfoNCE gradient's own cancellation)
    cos_head, head_name = min((_cos(gl_ref[k], p.grad.cpu()), k) for k, p in m.lang_rew.named_parameters() if p.numel() > 1)
    print(f"configs[2] r{size}: encoder gradient cosine vs checker {cos_all:.5f} (worst conv tensor {worst:.5f}), |g| ratio {(nb / tot) ** 0.5:.4f}; "
          f"reward-head worst tensor cosine {cos_head:.6f} ({head_name})")
    assert cos_all >= 0.99 and worst >= 0.97 and 0.97 <= (nb / tot) ** 0.5 <= 1.03
    # the head sees embeddings that differ by one bf16 rounding sequence (3-4e-3 l2-rel) and its gradient is a difference of
    # InfoNCE terms: measured 0.979 (ResNet-18, D = 512) / 0.987 (ResNet-50, D = 2048) on the worst of its ten tensors
    assert cos_head >= 0.95
    # Adam consumed both owners' gradients: every encoder weight with a non-negligible gradient moved by ~lr in its direction
    moved = m.convnet.flat_params() - before
    assert float(moved.abs().max()) <= 1.01e-4 and float((moved != 0).float().mean()) > 0.95
    assert m.encoder_opt._steps == [1, 1]


def test_config3_full_loss_fp32_step_runs_through_ddp_wrapper_protocol(hip):
    """BASELINE configs[3] on one GPU: the full R3M loss (LP + TCN + language) in fp32 with the stage hooks firing in backward
    order and covering the flat gradient buffer exactly once — the protocol DistributedR3M relies on (the collectives themselves:
    tests/test_gpu_ddp.py)."""
    from oracle import detgen
    from r3m_amd import R3M
    from r3m_amd.parallel import SingleDevice
    from r3m_amd.trainer import Trainer
    B = 2
    m = R3M("cuda", 1e-4, 1024, size=50, l2weight=1e-5, l1weight=1e-5, langweight=1.0, tcnweight=1.0)
    model = SingleDevice(m).to(DEV)
    seen = []
    m.convnet._stage_hook = lambda stage, off, cnt: seen.append((stage, off, cnt, m.lang_rew.has_grads()))
    frames = torch.from_numpy(detgen.frames("cfg3", (B, 5, 3, 224, 224))).to(DEV)
    feats = torch.from_numpy(detgen.uniform("langfeat", (B, 768), -0.6, 0.6)).to(DEV)
    metrics, _ = Trainer(1).update(model, (frames, feats), 0)
    assert all(np.isfinite(v) for v in metrics.values()) and "rewloss" in metrics and "tcnloss" in metrics
    assert [s[0] for s in seen] == [0, 1, 2, 3]
    assert all(s[3] for s in seen)                     # the reward head's backward is complete before the first encoder stage ends
    spans = sorted((off, off + cnt) for _, off, cnt, _ in seen)
    assert spans[0][0] == 0 and spans[-1][1] == m.convnet.flat_params().numel()
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    # later stages own EARLIER parameters (backward order): stage 0 = layer4 = the tail of the flat buffer
    assert seen[0][1] > seen[3][1]


def test_config4_resnet34_bf16_rctraj_full_size(hip):
    """BASELINE configs[4] at its own size: ResNet-34, 512 clips = 2560 frames, bf16, `rctraj` RandomResizedCrop on the GPU from
    resident uint8 256x256 clips, inside Trainer.update. Oracle-free properties tie it to the pinned small cases:
      * full-frame boxes on 224x224 clips == the no-crop path, bit for bit (crop kernel is the identity there);
      * eval-mode rows of the 2560-frame plan == the same (cropped) frames through an 8-frame plan (one bf16 rounding);
      * a full training step with random `rctraj` boxes runs, every metric finite, every parameter gradient finite and non-zero;
      * `rctraj` = ONE box per clip: the 5 frames of a clip share their box (identical frames stay identical after the crop)."""
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs the 288 GB of an MI355X")
    from r3m_amd import R3M, augment
    from r3m_amd.parallel import SingleDevice
    from r3m_amd.trainer import Trainer
    B = 512
    torch.manual_seed(5)
    m = R3M("cuda", 1e-4, 1024, size=34, l2weight=1e-5, l1weight=1e-5, langweight=0.0, tcnweight=1.0, precision="bf16").to(DEV)
    net = SingleDevice(m)
    g = torch.Generator(device=DEV).manual_seed(10)

    # ---- identity boxes == no crop ----
    raw224 = torch.randint(0, 256, (B, 5, 3, 224, 224), generator=g, device=DEV, dtype=torch.int32).to(torch.uint8)
    full_boxes = torch.tensor([[0, 0, 224, 224]], dtype=torch.int32).repeat(B, 1)
    same = augment.crop_resize(raw224.reshape(B * 5, 3, 224, 224), full_boxes, 5)
    assert torch.equal(same, raw224.reshape(B * 5, 3, 224, 224).float())
    m.train()
    with torch.no_grad():                                         # calibrate the running statistics for the eval-mode check
        m(same[:64])
        m(same[64:128])
    m.eval()
    with torch.no_grad():
        h_crop = m(same).clone()
        h_plain = m(raw224.reshape(B * 5, 3, 224, 224).float()).clone()
    assert torch.equal(h_crop, h_plain)
    del raw224, same, h_plain

    # ---- random rctraj boxes from 256x256 uint8 clips ----
    raw = torch.randint(0, 256, (B, 5, 3, 256, 256), generator=g, device=DEV, dtype=torch.int32).to(torch.uint8)
    raw[7, 1:] = raw[7, :1]                                       # a clip of five identical frames
    box_gen = torch.Generator().manual_seed(99)
    x = augment.random_resized_crop(raw, per_clip=True, generator=box_gen)
    assert x.shape == (B, 5, 3, 224, 224) and x.dtype == torch.float32
    assert float(x.min()) >= 0.0 and float(x.max()) <= 255.0
    assert all(torch.equal(x[7, 0], x[7, t]) for t in range(1, 5))
    xr = x.reshape(B * 5, 3, 224, 224)
    with torch.no_grad():
        h_full = m(xr).clone()
        idx = torch.tensor([0, 1, 7, 128, 255, 640, 1000, 2559], device=DEV)
        h_small = m(xr[idx]).clone()
    e_max, _ = rel_err(h_full[idx].cpu().numpy(), h_small.cpu().numpy())
    print(f"configs[4] eval rows, 2560-frame vs 8-frame plan: max-rel {e_max:.3e}")
    assert torch.isfinite(h_full).all() and e_max <= 2.0 ** -7
    del x, xr, h_full

    # ---- the step itself, crop inside it (what bench.py --size 34 --clips-per-gpu 512 --precision bf16 --doaug rctraj times) ----
    tr = Trainer(eval_freq=10 ** 9)
    losses = []
    for it in range(2):
        torch.manual_seed(100 + it)
        frames = augment.random_resized_crop(raw, per_clip=True, generator=box_gen, fused=True)   # boxes only: crop in the stem pre-pass
        assert isinstance(frames, augment.CroppedClips) and frames.shape == (B, 5, 3, 224, 224)
        met, _ = tr.update(net, (frames, [""] * B), it)
        assert all(np.isfinite(v) for v in met.values()), met
        losses.append(met["full_loss"])
    P = dict(m.convnet.named_parameters())
    gflat = m.convnet.flat_grads()
    assert torch.isfinite(gflat).all() and all(float(p.grad.abs().max()) > 0 for p in P.values())
    print(f"configs[4] two steps at 2560 frames: full_loss {losses}")
    del raw, frames
    torch.cuda.empty_cache()
