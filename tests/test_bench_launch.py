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
    assert names == ["configs[2]", "configs[3]", "configs[4]", "encoder_only_256_frames"]
    lab = {n: bench.workload_label(dict({"encoder_only_frames": 0}, **w), 1) for n, w in bench.SECONDARY}
    for n in names[:3]:                                    # each secondary BASELINE workload is labelled as the config it is
        assert lab[n].startswith("BASELINE " + n), (n, lab[n])
    assert lab["encoder_only_256_frames"] == "encoder-only continuity point"      # (the literal "bs=256/GPU" reading)
    assert set(bench.FWD_GFLOP_PER_FRAME) == {18, 34, 50} and abs(3 * bench.FWD_GFLOP_PER_FRAME[50] - 24.2868) < 0.3
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


def test_gpu_count_error_is_raised_before_the_process_group():
    assert bench.gpu_count_error(8, 8, 7, 8, False) is None
    msg = bench.gpu_count_error(8, 8, 5, 4, False)
    assert "LOCAL_RANK=5" in msg and "4 GPU(s) visible" in msg and "--gpus 4" in msg
    assert "no GPU visible" in bench.gpu_count_error(1, 1, 0, 0, False)
    assert bench.gpu_count_error(2, 2, 1, 1, True) is None          # --share-gpu: two ranks on the one device


def test_rank_cpu_plan_follows_the_gpu_numa_node():
    """N > 1 host placement (r3m_amd/utils/affinity.py): ranks whose GPUs hang off one NUMA node split THAT node's cores into
    disjoint slices; unknown topology falls back to an even split of the process's CPU set."""
    from r3m_amd.utils.affinity import parse_cpulist, plan_rank_cpus
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and parse_cpulist("") == [] and parse_cpulist("5") == [5]
    # two sockets x 64 cores (+ SMT siblings 128..255), GPUs 0-3 on node 0, 4-7 on node 1 — the MI355X 8-GPU host layout
    node_cpus = {0: parse_cpulist("0-63,128-191"), 1: parse_cpulist("64-127,192-255")}
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    allowed = list(range(256))
    plans = [plan_rank_cpus(r, nodes, node_cpus, allowed, max_cpus=16) for r in range(8)]
    for r, p in enumerate(plans):
        assert len(p) == 16 and set(p) <= set(node_cpus[nodes[r]]), (r, p)
    for a in range(8):
        for b in range(a + 1, 8):
            assert not set(plans[a]) & set(plans[b]), (a, b)          # disjoint: no rank's launcher shares a core with another's
    assert plans[0][0] == 0 and plans[4][0] == 64
    # cgroup mask smaller than the node: only allowed cores are used
    p = plan_rank_cpus(1, [0, 0], {0: list(range(64))}, list(range(8)), max_cpus=16)
    assert p == [4, 5, 6, 7]
    # unknown topology (no sysfs node, or numa_node = -1): even split of the allowed set by local rank
    plans = [plan_rank_cpus(r, [None] * 4, {}, list(range(32)), max_cpus=16) for r in range(4)]
    assert plans == [list(range(0, 8)), list(range(8, 16)), list(range(16, 24)), list(range(24, 32))]
    # more ranks than cores: nobody gets an empty mask
    assert all(plan_rank_cpus(r, [None] * 8, {}, [0, 1], max_cpus=4) for r in range(8))


def test_bind_rank_never_raises_and_keeps_a_nonempty_mask():
    from r3m_amd.utils import affinity
    before = os.sched_getaffinity(0)
    try:
        info = affinity.bind_rank(1, 2, device_indices=[0, 1], set_threads=False)    # no GPU here: topology unknown -> even split
        assert "error" in info or len(os.sched_getaffinity(0)) >= 1
        if "error" not in info:
            assert set(os.sched_getaffinity(0)) <= set(before) and info["cpus"]
    finally:
        os.sched_setaffinity(0, before)
    # a slice smaller than what the caller needs (launcher + loader workers) leaves the mask alone and says so
    info = affinity.bind_rank(0, 2, device_indices=[0, 1], set_threads=False, min_cpus=10 ** 6)
    assert ("skipped" in info or "error" in info) and os.sched_getaffinity(0) == before
