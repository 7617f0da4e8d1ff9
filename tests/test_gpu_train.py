"""-m gpu: the drop-in training loop (r3m_amd/train_representation.py) end to end on synthetic clips: metrics logged, snapshot in
the reference's layout ({'r3m': module.-prefixed state dict, 'global_step'}; train_representation.py:123-138), resume, and
the on-GPU rctraj crop in the loop."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_loop_snapshot_resume(hip, tmp_path, monkeypatch):
    from r3m_amd import load_r3m, train_representation as tr
    monkeypatch.chdir(tmp_path)
    args = ["dataset=synthetic", "batch_size=2", "train_steps=3", "eval_freq=2", "num_workers=0", "agent.size=18", "doaug=rctraj",
            "experiment=t1"]
    tr.main(args)
    out = tmp_path / "r3moutput" / "t1"
    snap = torch.load(out / "snapshot.pt", map_location="cpu")
    assert snap["global_step"] == 2 and "encoder_opt" in snap
    keys = list(snap["r3m"].keys())
    assert len(keys) == 120 and keys[0] == "module.convnet.conv1.weight" and tuple(snap["r3m"][keys[0]].shape) == (64, 3, 7, 7)
    assert (out / "snapshot_0.pt").exists() and (out / "snapshot_2.pt").exists()
    recs = [json.loads(l) for l in open(out / "logs_rank0" / "metrics.jsonl")]
    train = [r for r in recs if r["ty"] == "train"]
    assert len(train) == 3 and all(k in train[0] for k in ("l2loss", "l1loss", "l0loss", "tcnloss", "aligned", "full_loss"))
    assert all(torch.isfinite(torch.tensor(r["full_loss"])) for r in recs)
    # resume: picks up global_step and continues to train_steps
    tr.main(["dataset=synthetic", "batch_size=2", "train_steps=4", "eval_freq=100", "num_workers=0", "agent.size=18", "experiment=t1"])
    recs2 = [json.loads(l) for l in open(out / "logs_rank0" / "metrics.jsonl")]
    assert [r["step"] for r in recs2 if r["ty"] == "train"][-2:] == [2, 3]
    # the snapshot loads through the inference API's state-dict contract
    monkeypatch.setenv("HOME", str(tmp_path))
    d = tmp_path / ".r3m" / "r3m_18"
    d.mkdir(parents=True)
    torch.save({"r3m": snap["r3m"]}, d / "model.pt")
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (d / "config.yaml").write_text(open(os.path.join(here, "r3m_amd", "cfgs", "config_rep.yaml")).read().replace("size: 34", "size: 18"))
    rep = load_r3m("resnet18")
    rep.eval()
    rep = rep.to("cuda:0")
    x = torch.randint(0, 256, (3, 3, 224, 224), device="cuda:0").float()
    with torch.no_grad():
        h = rep(x)
    assert tuple(h.shape) == (3, 512) and torch.isfinite(h).all() and (h >= 0).all()
