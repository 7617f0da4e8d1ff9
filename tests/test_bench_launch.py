"""bench.py's launch logic (CPU): `python bench.py --gpus N` must produce N ranks by itself — the reference scales inside
one process (nn.DataParallel, /root/reference/r3m/train_representation.py:27-31), so nobody types a launcher there — while the
driver's `python -m torch.distributed.run ... bench.py --gpus N` form keeps working."""
import json
import os
import sys

import pytest

import bench


def _args(*argv):
    return bench.parse_args(list(argv))


def test_single_gpu_runs_in_process():
    assert bench.launch_plan(_args(), {}, []) is None
    assert bench.launch_plan(_args("--gpus", "1", "--steps", "3"), {"HOME": "/root"}, ["--gpus", "1", "--steps", "3"]) is None


def test_multi_gpu_without_launcher_self_spawns():
    argv = ["--gpus", "8", "--steps", "7", "--warmup", "2"]
    cmd = bench.launch_plan(_args(*argv), {}, argv, port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[script + 1:] == argv                       # the user's flags reach every rank unchanged


def test_rank_process_does_not_respawn():
    env = {"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "8", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"}
    assert bench.launch_plan(_args("--gpus", "8"), env, ["--gpus", "8"]) is None
    one = {"RANK": "0", "WORLD_SIZE": "1", "MASTER_PORT": "1"}
    assert bench.launch_plan(_args("--gpus", "1"), one, ["--gpus", "1"]) is None
    assert bench.launch_plan(_args("--gpus", "1", "--force-launcher"), one, []) is None


def test_world_size_mismatch_is_an_error():
    with pytest.raises(SystemExit) as e:
        bench.launch_plan(_args("--gpus", "4"), {"RANK": "0", "WORLD_SIZE": "2"}, ["--gpus", "4"])
    assert "WORLD_SIZE=2" in str(e.value)
    with pytest.raises(SystemExit):
        bench.launch_plan(_args("--gpus", "0"), {}, [])


def test_force_launcher_on_one_gpu():
    cmd = bench.launch_plan(_args("--gpus", "1", "--force-launcher"), {}, ["--gpus", "1", "--force-launcher"], port=5)
    assert "--nproc-per-node=1" in cmd


def test_free_port_is_bindable():
    import socket
    p = bench.free_port()
    with socket.socket() as s:
        s.bind(("127.0.0.1", p))


def test_headline_detection_and_secondary_set():
    assert bench.is_headline(_args())
    assert bench.is_headline(_args("--gpus", "8", "--steps", "3"))
    for flags in (["--precision", "bf16"], ["--size", "34"], ["--langweight", "1"], ["--doaug", "rctraj"],
                  ["--encoder-only-frames", "256"], ["--clips-per-gpu", "8"]):
        assert not bench.is_headline(_args(*flags)), flags
    names = [n for n, _ in bench.SECONDARY]
    assert names == ["configs[2]", "configs[3]", "configs[4]"]
    lab = {n: bench.workload_label(dict(w, encoder_only_frames=0), 1) for n, w in bench.SECONDARY}
    for n in names:                                        # each secondary workload is labelled as the config it is
        assert lab[n].startswith("BASELINE " + n), (n, lab[n])
    assert bench.workload_label(dict(size=50, precision="fp32", langweight=0.0, doaug="none", encoder_only_frames=0), 8).startswith(
        "BASELINE configs[1], replicated on 8 GPUs")


def test_pmc_summary_is_only_used_for_its_own_workload_and_kernel():
    """roofline.traffic comes from a committed counter summary; it must name a file that exists, the workload
    it was collected on and the kernel class it describes, or it is withheld."""
    for name, bf16 in (("pmc_latest.json", False), ("pmc_latest_bf16.json", True)):
        pj = json.load(open(os.path.join(bench.ROOT, "profiles", name)))
        assert os.path.exists(os.path.join(bench.ROOT, pj["source"])), pj["source"]
        w = pj["workload"]
        t, src = bench.pmc_traffic(bf16, w["size"], w["clips"], bench.KCLASS[0])
        assert t == pj["dominant_kernel_hbm_bytes_per_launch"] and "not measured live" in src
        t, src = bench.pmc_traffic(bf16, 34 if w["size"] != 34 else 18, w["clips"], bench.KCLASS[0])
        assert t is None and "not this workload" in src
        t, src = bench.pmc_traffic(bf16, w["size"], w["clips"], bench.KCLASS[3])
        assert t is None and "dominant class here" in src


def test_gloo_share_gpu_flags_reach_the_ranks():
    argv = ["--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "2"]
    a = _args(*argv)
    assert a.backend == "gloo" and a.share_gpu
    cmd = bench.launch_plan(a, {}, argv, port=7)
    assert cmd[-len(argv):] == argv and "--nproc-per-node=2" in cmd
    assert _args().backend == "nccl" and not _args().share_gpu
