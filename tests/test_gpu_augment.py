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


@pytest.mark.parametrize("hw", [(500, 500), (240, 427), (600, 300), (256, 256), (224, 300)])
@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32])
def test_resize_center_crop_branch_matches_oracle(hip, hw, dtype):
    """R3M.forward for inputs that are not 224x224 (/root/reference/r3m/models/models_r3m.py:85-90: Resize(256) + CenterCrop(224)
    on x/255; example.py feeds a 500x500 image): the one-pass HIP gather (csrc/augment.hip) against the oracle's restatement of
    the torchvision transforms (oracle/r3m_ref.resize_center_crop_ref), incl. non-square frames in both orientations."""
    from oracle import r3m_ref
    from r3m_amd import augment
    H, W = hw
    g = torch.Generator().manual_seed(H * 1000 + W)
    x = torch.randint(0, 256, (3, 3, H, W), generator=g, dtype=torch.uint8)
    ref = r3m_ref.resize_center_crop_ref(x)
    assert ref.shape == (3, 3, 224, 224)
    out = augment.resize_center_crop(x.to("cuda:0").to(dtype)).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-4)


def test_r3m_forward_takes_the_resize_branch(hip):
    """model(obs, obs_shape=[3,500,500]) == model(Resize+CenterCrop(obs)) — the public forward with the reference's signature."""
    import numpy as np
    from oracle import detgen, r3m_ref
    from r3m_amd import R3M
    m = R3M("cuda", 1e-4, 1024, size=18, langweight=0.0, tcnweight=1.0)
    shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
    m.convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict(shapes).items()})
    m = m.to("cuda:0").eval()
    x = torch.from_numpy(detgen.frames("big", (2, 3, 500, 500)))
    with torch.no_grad():
        h_branch = m(x.to("cuda:0"), obs_shape=[3, 500, 500]).cpu()
        h_direct = m(r3m_ref.resize_center_crop_ref(x).to("cuda:0")).cpu()
    assert h_branch.shape == (2, 512)
    assert float((h_branch - h_direct).abs().max() / h_direct.abs().max()) < 1e-4


def test_vectorised_box_sampler_matches_scalar_algorithm(hip):
    """sample_boxes draws every box of a batch in a few tensor ops; its boxes must be the ones torchvision's get_params loop
    (restated scalar form, r3m_amd.augment._sample_boxes_scalar) picks from the SAME per-try random numbers."""
    from r3m_amd import augment
    for (H, W) in ((256, 256), (224, 300), (100, 400)):
        g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
        a = augment.sample_boxes(257, H, W, generator=g1)
        b = augment._sample_boxes_scalar(257, H, W, generator=g2)
        assert torch.equal(a, b), (H, W)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("per_clip,dtype", [(True, torch.uint8), (False, torch.uint8), (True, torch.float32)])
def test_crop_inside_the_stem_prepass_equals_crop_then_forward(hip, precision, per_clip, dtype):
    """SURVEY.md §8(f)1 as written: ONE gather-bilinear pass from the uint8 clips into the normalised stem image
    (r3m_resnet_forward_crop via augment.CroppedClips). Must give the SAME embeddings and parameter gradients as the stand-alone
    crop kernel followed by the ordinary forward — both run the same float operations (csrc/augment_dev.h) — for rctraj (one box
    per clip) and rc (one per frame), uint8 and float clips, fp32 and bf16 plans, train and eval mode."""
    import numpy as np
    from oracle import detgen
    from r3m_amd import R3M, augment
    B, T, H, W = 3, 5, 256, 320
    g = torch.Generator().manual_seed(21)
    raw = torch.randint(0, 256, (B, T, 3, H, W), generator=g, dtype=torch.uint8).to("cuda:0").to(dtype)
    boxes = augment.sample_boxes(B if per_clip else B * T, H, W, generator=g)
    boxes[0] = torch.tensor([H - 9, W - 6, 9, 6], dtype=torch.int32)       # tiny corner box: up-sampling + edge clamps
    clips = augment.CroppedClips(raw, boxes, T if per_clip else 1)
    assert clips.shape == (B, T, 3, 224, 224) and clips.reshape(B * T, 3, 224, 224).shape == (B * T, 3, 224, 224)
    pixels = clips.materialize()
    assert pixels.shape == (B, T, 3, 224, 224)
    m = R3M("cuda", 1e-4, 1024, size=18, langweight=0.0, tcnweight=1.0, precision=precision)
    shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
    m.convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict(shapes).items()})
    m = m.to("cuda:0")
    for training in (False, True):
        m.train(training)
        res = []
        for inp in (pixels.reshape(B * T, 3, 224, 224), clips.reshape(B * T, 3, 224, 224)):
            m.encoder_opt.zero_grad()
            h = m(inp)
            (h * torch.linspace(0.5, 1.5, h.shape[1], device=h.device)).sum().backward()
            res.append((h.detach().clone(), m.convnet.flat_grads().clone()))
        assert torch.equal(res[0][0], res[1][0]), f"embeddings differ ({precision}, training={training})"
        assert torch.equal(res[0][1], res[1][1]), f"gradients differ ({precision}, training={training})"
    with pytest.raises(ValueError):
        clips.reshape(B * T * 3, 224, 224)


@pytest.mark.parametrize("H,W", [(256, 256), (256, 320), (300, 512), (240, 700), (9, 4), (33, 7)])
def test_u8_stem_prepass_crop_is_bit_identical_to_the_float_source_kernel(hip, H, W):
    """Round 6: the uint8 crop pre-pass of the bf16 stem (csrc/stem_bf16.hip stem_prep16_crop_u8_kernel) fetches its source bytes as
    unaligned dword loads shared by a pixel's two taps and, for down-sampling factors up to 1.5, by a pixel pair — clamped into the
    source row. Same float operations as the generic kernel on float clips (augment_dev.h): the padded bf16 images must be identical
    BIT FOR BIT — boxes that touch every edge (clamped loads), tiny boxes (up-sampling: both taps of many pixels in one byte pair),
    wide clips (factors > 1.5: unshared loads), rows of 4-7 bytes."""
    g = torch.Generator().manual_seed(5)
    NF = 7
    raw = torch.randint(0, 256, (NF, 3, H, W), generator=g, dtype=torch.uint8)
    boxes = torch.tensor([[0, 0, H, W],                                   # the whole clip: strongest down-sampling this clip allows
                          [H - min(H, 5), W - min(W, 4), min(H, 5), min(W, 4)],   # bottom-right corner, 4 columns: every load clamped
                          [0, W - min(W, 9), min(H, 6), min(W, 9)],       # top-right strip
                          [H // 3, 0, max(1, H // 2), max(1, W // 2)],    # left edge
                          [1, 1, 1, 1],                                   # one source pixel
                          [H // 5, W // 7, max(2, H // 2), max(2, int(W * 0.6))],
                          [0, max(0, W - 224), min(H, 224), min(W, 224)]], dtype=torch.int32)
    raw_d = raw.to("cuda:0")
    rawf_d = raw_d.float()
    boxes_d = boxes.to("cuda:0")
    nbytes = hip.r3m_stem_xn16_bytes(NF)
    out_u8 = torch.full((nbytes,), 0x5a, dtype=torch.uint8, device="cuda:0")
    out_f = torch.full((nbytes,), 0xa5, dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    assert hip.r3m_stem_prep_crop(raw_d.data_ptr(), 1, boxes_d.data_ptr(), 1, H, W, out_u8.data_ptr(), NF, 1, st) == 0, hip.r3m_last_error()
    assert hip.r3m_stem_prep_crop(rawf_d.data_ptr(), 0, boxes_d.data_ptr(), 1, H, W, out_f.data_ptr(), NF, 1, st) == 0, hip.r3m_last_error()
    torch.cuda.synchronize()
    a = out_u8.view(torch.int16).view(NF, 232, 704)
    b = out_f.view(torch.int16).view(NF, 232, 704)
    bad = (a != b).nonzero()
    assert bad.numel() == 0, (bad[:5].tolist(), int((a != b).sum()))
    # and the image is not trivially empty: the interior rows carry values
    assert int((a[:, 3:227, 9:681] != 0).sum()) > 0.9 * NF * 224 * 672
