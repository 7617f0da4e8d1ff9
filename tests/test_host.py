"""not gpu: host-side logic of the drop-in surface — config resolver, cleanup_config, load_r3m contract, state-dict key set
and flat-buffer views of the encoder module, loud failure on CPU tensors, deterministic generator."""
import copy
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_defaults_and_overrides():
    from r3m_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, "r3m_amd", "cfgs", "config_rep.yaml"))
    # reference defaults: r3m/cfgs/config_rep.yaml:8-41
    assert cfg.batch_size == 32 and cfg.train_steps == 2000000 and cfg.eval_freq == 20000 and cfg.seed == 1
    assert cfg.lr == pytest.approx(1e-4) and isinstance(cfg.lr, float)
    assert cfg.agent.lr == cfg.lr and cfg.agent.bs == 32 and cfg.agent.device == "cuda"      # ${...} interpolation
    assert cfg.agent.size == 34 and cfg.agent.l2dist is True and cfg.agent.tcnweight == 1.0 and cfg.agent.langweight == 0.0
    assert cfg.agent.l1weight == pytest.approx(1e-5) and cfg.agent["_target_"] == "r3m.R3M"
    cfg = load_config(os.path.join(ROOT, "r3m_amd", "cfgs", "config_rep.yaml"), ["batch_size=16", "agent.size=50", "lr=3e-4",
                                                                                 "doaug=rctraj", "agent.langweight=1.0"])
    assert cfg.agent.bs == 16 and cfg.agent.size == 50 and cfg.agent.lr == pytest.approx(3e-4) and cfg.doaug == "rctraj"


def test_cleanup_config_and_language_head_removal():
    import r3m_amd
    from r3m_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, "r3m_amd", "cfgs", "config_rep.yaml"), ["agent.langweight=1.0", "agent.size=18"])
    cfg.agent["extra_key"] = 3
    clean = r3m_amd.cleanup_config(cfg)
    assert set(clean.keys()) <= set(r3m_amd.VALID_ARGS) and "extra_key" not in clean
    assert clean["langweight"] == 0 and clean["_target_"] == "r3m.R3M"
    assert "extra_key" in cfg.agent                      # input untouched (deepcopy), like the reference
    sd = {"module.convnet.conv1.weight": 1, "module.lang_rew.pred.0.weight": 2, "module.lang_enc.model.x": 3}
    assert list(r3m_amd.remove_language_head(sd).keys()) == ["module.convnet.conv1.weight"]


def test_load_r3m_contract(tmp_path, monkeypatch):
    import r3m_amd
    with pytest.raises(NameError, match="Invalid Model ID"):
        r3m_amd.load_r3m("resnet101")
    monkeypatch.setenv("HOME", str(tmp_path))
    with pytest.warns(RuntimeWarning):
        rep = r3m_amd.load_r3m("resnet18")
    assert hasattr(rep, "module") and isinstance(rep.module, r3m_amd.R3M)
    assert rep.module.langweight == 0 and rep.module.outdim == 512 and rep.module.size == 18
    keys = list(rep.state_dict().keys())
    assert len(keys) == 120 and all(k.startswith("module.convnet.") for k in keys)
    # a checkpoint saved in the reference layout round-trips (train_representation.py:123-138 / __init__.py:73-74)
    d = tmp_path / ".r3m" / "r3m_18"
    d.mkdir(parents=True)
    sd = {k: torch.full_like(v, 0.25) if v.dtype.is_floating_point else v for k, v in rep.state_dict().items()}
    sd["module.lang_rew.pred.0.weight"] = torch.zeros(3)
    torch.save({"r3m": sd}, d / "model.pt")
    (d / "config.yaml").write_text(open(os.path.join(ROOT, "r3m_amd", "cfgs", "config_rep.yaml")).read().replace("size: 34", "size: 18"))
    rep2 = r3m_amd.load_r3m("resnet18")
    assert float(rep2.module.convnet.conv1.weight.mean()) == 0.25
    import r3m
    assert r3m.load_r3m is r3m_amd.load_r3m and r3m.R3M is r3m_amd.R3M


@pytest.mark.parametrize("size,nkeys,nparams", [(18, 120, 11176512), (34, 216, 21284672), (50, 318, 23508032)])
def test_state_dict_matches_torchvision_layout(size, nkeys, nparams):
    from oracle import resnet_ref
    from r3m_amd.encoder import HipResNet
    enc = HipResNet(size)
    ref = {18: resnet_ref.resnet18, 34: resnet_ref.resnet34, 50: resnet_ref.resnet50}[size]()
    ref.fc = torch.nn.Identity()
    sd, rsd = enc.state_dict(), ref.state_dict()
    assert list(sd.keys()) == list(rsd.keys()) and len(sd) == nkeys
    for k in sd:
        assert tuple(sd[k].shape) == tuple(rsd[k].shape) and sd[k].dtype == rsd[k].dtype, k
    assert sum(p.numel() for p in enc.parameters()) == nparams
    assert [n for n, _ in enc.named_parameters()] == [n for n, _ in ref.named_parameters()]
    # every tensor is a view of the flat buffers; conv weights are physically OHWI
    assert enc._is_flat()
    w = enc.layer1[0].conv1.weight if False else dict(enc.named_parameters())["layer1.0.conv1.weight"]
    assert w.permute(0, 2, 3, 1).is_contiguous()
    # load_state_dict from an OIHW-contiguous reference checkpoint keeps the views and the values
    enc.load_state_dict(rsd)
    assert enc._is_flat()
    assert torch.equal(dict(enc.named_parameters())["layer1.0.conv1.weight"], rsd["layer1.0.conv1.weight"])
    # deepcopy / .to() re-flatten
    enc2 = copy.deepcopy(enc)
    enc2._ensure()
    assert enc2._is_flat() and enc2.flat_params().data_ptr() != enc.flat_params().data_ptr()
    assert torch.equal(enc2.state_dict()["layer1.0.conv1.weight"], rsd["layer1.0.conv1.weight"])
    enc2 = enc2.to(torch.device("cpu"))
    assert enc2._is_flat()


def test_no_cpu_fallback():
    from r3m_amd import R3M
    m = R3M("cpu", 1e-4, 1024, size=18, langweight=0.0, tcnweight=1.0)
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.zeros(2, 3, 224, 224))
    with pytest.raises(ValueError):
        R3M("cpu", 1e-4, 1024, size=0)
    for attr in ("l2weight", "l1weight", "tcnweight", "langweight", "l2dist", "size", "num_negatives", "outdim", "encoder_opt"):
        assert hasattr(m, attr)
    assert m.num_negatives == 3 and m.convnet.training


def test_detgen_is_stable():
    from oracle import detgen
    a = detgen.uniform("x", (5,), -1, 1)
    np.testing.assert_allclose(a, detgen.uniform("x", (5,), -1, 1))
    assert a.dtype == np.float32 and np.all(np.abs(a) <= 1)
    f = detgen.frames("f", (1000,))
    assert f.min() >= 0 and f.max() <= 255 and np.all(f == np.floor(f))
    p = detgen.permutation("p", 17)
    assert sorted(p.tolist()) == list(range(17))
    # pinned values: the generator must never drift (golden inputs are regenerated from it on the GPU box)
    np.testing.assert_allclose(detgen.unit("pin", 3), [0.2076382040977478, 0.2765554189682007, 0.4500434398651123], rtol=0, atol=1e-7)


def test_precision_option_is_plumbed():
    """`precision` is an extra constructor key on top of the reference's (models_r3m.py:17-19): default fp32, validated,
    reachable from the config as agent.precision=bf16, and never changes the parameter / state-dict surface."""
    import os
    from r3m_amd import R3M, config
    from r3m_amd.encoder import HipResNet
    with pytest.raises(ValueError):
        HipResNet(18, precision="fp16")
    m32 = R3M("cpu", 1e-4, 64, size=18, langweight=0.0, tcnweight=1.0)
    m16 = R3M("cpu", 1e-4, 64, size=18, langweight=0.0, tcnweight=1.0, precision="bf16")
    assert m32.convnet.precision == "fp32" and m16.convnet.precision == "bf16"
    sd32, sd16 = m32.state_dict(), m16.state_dict()
    assert list(sd32.keys()) == list(sd16.keys())
    assert all(sd16[k].dtype == sd32[k].dtype and sd16[k].shape == sd32[k].shape for k in sd32)   # fp32 masters in both
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.load_config(os.path.join(root, "r3m_amd", "cfgs", "config_rep.yaml"), ["agent.precision=bf16"])
    assert cfg.agent.precision == "bf16"


def test_deepcopy_and_pickle_drop_native_handles():
    """copy.deepcopy(R3M) must not duplicate native plan handles / the arena, and the optimizer copy must keep the
    owners protocol, its per-owner step counters and moments (the reference R3M + torch.optim.Adam deep-copy cleanly)."""
    import pickle
    from r3m_amd import R3M
    m = R3M("cpu", 1e-4, 64, size=18, langweight=1.0, tcnweight=1.0)
    enc = m.convnet
    enc._plans = {8: 0xdeadbeef}                       # what a GPU forward would have left behind
    enc._arena = torch.zeros(16, dtype=torch.uint8)
    enc.flat_grads()
    enc._stage_hook = lambda *a: None
    m.encoder_opt._steps = [3, 1]
    m.encoder_opt._m[0] = torch.full_like(enc.flat_params(), 0.5)
    m.encoder_opt._v[0] = torch.full_like(enc.flat_params(), 0.25)
    try:
        for m2 in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
            e2 = m2.convnet
            assert e2._plans == {} and e2._arena is None and e2._flat_g is None and e2._stage_hook is None
            assert all(p.grad is None for p in e2.parameters())
            e2._ensure()
            assert e2._is_flat() and e2.flat_params().data_ptr() != enc.flat_params().data_ptr()
            assert torch.equal(e2.flat_params(), enc.flat_params())
            o2 = m2.encoder_opt
            assert o2.owners[0] is e2 and o2.owners[1] is m2.lang_rew          # owners follow the copy, not the original
            assert o2._steps == [3, 1] and torch.equal(o2._m[0], m.encoder_opt._m[0]) and o2._m[1] is None
            assert o2._m[0].data_ptr() != m.encoder_opt._m[0].data_ptr()
            assert {id(p) for g in o2.param_groups for p in g["params"]} == {id(p) for p in m2.parameters()}
        assert enc._plans == {8: 0xdeadbeef}           # the original keeps its own handles
    finally:
        enc._plans = {}                                # fake handle: never hand it to r3m_resnet_destroy


def test_adam_step_is_per_owner_and_persisted():
    """torch.optim.Adam keeps `step` per parameter — an owner without gradients (language head before its first
    backward) must not advance its bias correction; state_dict carries the per-owner counters and reads round-1 files."""
    from r3m_amd import R3M
    m = R3M("cpu", 1e-4, 64, size=18, langweight=1.0, tcnweight=1.0)
    opt = m.encoder_opt
    assert opt._steps == [0, 0]
    sd = opt.state_dict()
    assert sd["steps"] == [0, 0] and sd["step"] == 0
    opt.load_state_dict({**sd, "steps": [5, 2]})
    assert opt._steps == [5, 2] and opt._step == 5
    legacy = {k: v for k, v in sd.items() if k != "steps"}
    legacy["step"] = 7
    opt.load_state_dict(legacy)
    assert opt._steps == [7, 7]


def test_snapshot_filter_drops_only_frozen_text_keys():
    from r3m_amd.train_representation import filter_frozen_text_keys
    sd = {"module.convnet.conv1.weight": 1, "module.lang_rew.pred.0.weight": 2, "module.lang_enc.model.embeddings.w": 3,
          "lang_enc.model.x": 4}
    assert list(filter_frozen_text_keys(sd)) == ["module.convnet.conv1.weight", "module.lang_rew.pred.0.weight"]


def test_step_predicates_and_timer():
    from r3m_amd.utils import utils
    until = utils.Until(3, 1)
    assert [until(s) for s in range(5)] == [True, True, True, False, False] and utils.Until(None)(10 ** 9)
    every = utils.Every(4, 1)
    assert [s for s in range(10) if every(s)] == [0, 4, 8] and not utils.Every(None)(0) and not utils.Every(0)(0)
    t = utils.Timer()
    lap, total = t.reset()
    assert 0 <= lap <= total and t.total_time() >= total
    utils.set_seed_everywhere(5)
    a = torch.rand(3)
    utils.set_seed_everywhere(5)
    assert torch.equal(a, torch.rand(3))


def test_head_without_gradient_is_skipped_like_torch_adam():
    """zero_grad() must leave the language head with NO gradient until a backward produces one — torch.optim.Adam skips
    parameters whose .grad is None, and the data-parallel wrapper must not all-reduce a stale buffer."""
    from r3m_amd import R3M
    m = R3M("cpu", 1e-4, 64, size=18, langweight=1.0, tcnweight=1.0)
    head = m.lang_rew
    assert not head.has_grads()
    head._has_grads = True                                  # what a backward through the head leaves behind
    m.encoder_opt.zero_grad()
    assert not head.has_grads() and head._grad_fresh


def test_optimizer_state_of_another_owner_set_is_refused():
    """a snapshot saved with langweight=0 (one flat buffer) resumed into a model with a language head (two), or the
    reverse, must fail clearly instead of mis-assigning moments / step counters."""
    from r3m_amd import R3M
    enc_only = R3M("cpu", 1e-4, 64, size=18, langweight=0.0, tcnweight=1.0)
    with_head = R3M("cpu", 1e-4, 64, size=18, langweight=1.0, tcnweight=1.0)
    sd1, sd2 = enc_only.encoder_opt.state_dict(), with_head.encoder_opt.state_dict()
    assert len(sd1["steps"]) == 1 and len(sd2["steps"]) == 2
    with pytest.raises(ValueError, match="different langweight"):
        with_head.encoder_opt.load_state_dict(sd1)
    with pytest.raises(ValueError, match="different langweight"):
        enc_only.encoder_opt.load_state_dict(sd2)
    with_head.encoder_opt.load_state_dict(sd2)              # the matching one loads
    bad = dict(sd1, exp_avg=[torch.zeros(8)], exp_avg_sq=[torch.zeros(8)])
    with pytest.raises(ValueError, match="elements"):
        enc_only.encoder_opt.load_state_dict(bad)
    old = {"step": 4, "exp_avg": [None], "exp_avg_sq": [None], "param_groups": sd1["param_groups"]}   # round-1 layout: one shared step
    enc_only.encoder_opt.load_state_dict(old)
    assert enc_only.encoder_opt._steps == [4]


def test_box_sampler_matches_the_scalar_get_params_loop():
    """The vectorised (numpy) RandomResizedCrop box sampler picks, from the SAME per-try random numbers, the boxes torchvision's
    get_params loop (restated scalar form) picks — /root/reference/r3m/utils/data_loaders.py:47-50 uses scale=(0.2, 1.0)."""
    from r3m_amd import augment
    for n, H, W in ((257, 256, 256), (64, 224, 300), (9, 8, 1000), (5, 1000, 8)):
        g1, g2 = torch.Generator().manual_seed(11), torch.Generator().manual_seed(11)
        a = augment.sample_boxes(n, H, W, generator=g1)
        b = augment._sample_boxes_scalar(n, H, W, generator=g2)
        assert a.dtype == torch.int32 and tuple(a.shape) == (n, 4) and torch.equal(a, b), (n, H, W)
        top, left, h, w = a.unbind(1)
        assert (top >= 0).all() and (left >= 0).all() and (top + h <= H).all() and (left + w <= W).all()


def test_upload_small_keeps_values_and_dtype_on_a_cpu_target():
    """_lib.upload_small (pinned staging for the per-step crop boxes / permutations) degrades to a plain conversion when the
    target is not a GPU; values and the requested dtype are kept (the GPU path is exercised by every Trainer.update test)."""
    from r3m_amd import _lib
    t = torch.arange(12, dtype=torch.int64).reshape(3, 4)
    out = _lib.upload_small(t, "cpu", torch.int32)
    assert out.dtype == torch.int32 and torch.equal(out.to(torch.int64), t)
    assert torch.equal(_lib.upload_small(t, torch.device("cpu")), t)


def test_pick_slot_keeps_live_forwards():
    """Slot choice of the encoder's arena ring: forwards without a backward never evict a forward that awaits one."""
    from r3m_amd.encoder import _pick_slot
    # one slot, nothing live: everything runs in slot 0; the cursor only moves for saved forwards
    assert _pick_slot([False], 0, True) == (0, 0)
    assert _pick_slot([False], 0, False) == (0, 0)
    # one slot, live: a saved forward evicts it (its backward then raises and names max_live_forwards), an unsaved one goes to scratch
    assert _pick_slot([True], 0, True) == (0, 0)
    assert _pick_slot([True], 0, False) == (-1, 0)
    # two slots: free slots first, in cursor order
    assert _pick_slot([False, False], 0, True) == (0, 1)
    assert _pick_slot([True, False], 1, True) == (1, 0)
    assert _pick_slot([True, False], 0, False) == (1, 0)       # inference call: takes the free slot, cursor untouched
    assert _pick_slot([True, True], 1, True) == (1, 0)         # all live: the oldest (cursor) goes
    assert _pick_slot([True, True], 1, False) == (-1, 1)
    assert _pick_slot([False, True], 1, True) == (0, 1)


def test_profile_tools_keep_kernel_names_of_anonymous_namespaces():
    """tools/rocpd_stats.py / rocpd_pmc.py shorten rocprofv3 kernel names for the committed tables. Round 6's kernel-row bf16 kernel lives
    in an anonymous namespace: its demangled name starts with `(anonymous namespace)::`, and cutting at the first `(` left an EMPTY name
    (a whole evidence set went out with "" rows). Both the demangled and the mangled form must come out as the plain template name."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mod in ("rocpd_stats", "rocpd_pmc"):
        spec = importlib.util.spec_from_file_location(mod, os.path.join(root, "tools", mod + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        assert m.short("void r3m::(anonymous namespace)::conv3x3_row_bf16_kernel<128, 1>(r3m::GatherGemmParams, int, int)") == \
            "conv3x3_row_bf16_kernel<128, 1>"
        assert m.short("void r3m::pw_gemm_kernel<128, 128, 2, 2, 1, false, false, false, false>(r3m::GatherGemmParams)") == \
            "pw_gemm_kernel<128, 128, 2, 2, 1, false, false, false, false>"
    spec = importlib.util.spec_from_file_location("rocpd_stats", os.path.join(root, "tools", "rocpd_stats.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.short("_ZN3r3m12_GLOBAL__N_123conv3x3_row_bf16_kernelILi128ELi1EEEvNS_16GatherGemmParamsEii") == "conv3x3_row_bf16_kernel<128, 1>"
