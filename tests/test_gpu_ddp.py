"""-m gpu: the data-parallel wrapper end to end with the REAL engine backward: two ranks share the one GPU of the test box
(gloo backend on CUDA tensors — RCCL refuses two ranks on one device), each runs the encoder on its half of the frames,
the staged gradient all-reduce runs from the backward stage hooks, and the averaged gradients must equal a single-process
run on the full batch. BatchNorm runs on running statistics and the loss is the LP term only, so the two computations
are mathematically identical ('N ranks x B/N clips reproduces 1 rank x B gradients with BN in eval mode')."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

# encoder gradient collectives per step (r3m_amd/parallel.py: a finished stage slice under 16 MB waits for the next stage and
# goes out merged): ResNet-18 layer4 | layer3 + layer2 + layer1/stem; ResNet-34 / 50 layer4 | layer3 | layer2 + layer1/stem
ENC_SLICES = {18: 2, 34: 3, 50: 3}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(size, dev="cuda:0"):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import detgen
    from r3m_amd import R3M
    m = R3M("cuda", 1e-4, 1024, size=size, l2weight=1.0, l1weight=0.5, langweight=0.0, tcnweight=0.0)
    shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
    m.convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict(shapes).items()})
    return m.to(dev)


def _lp_backward(model, frames):
    """LP loss on running-stat BatchNorm: full = mean||h||_2 + 0.5 mean||h||_1 over the frames."""
    core = model.module
    core.convnet.eval()
    core.encoder_opt.zero_grad()
    h = model(frames)
    loss = torch.linalg.norm(h, ord=2, dim=-1).mean() + 0.5 * torch.linalg.norm(h, ord=1, dim=-1).mean()
    loss.backward()
    model.finish_gradient_sync()
    torch.cuda.synchronize()
    return core.convnet.flat_grads().clone()


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import detgen
        from r3m_amd.parallel import DistributedR3M, make_network_wrapper
        m = _build(18)
        net = make_network_wrapper(m)
        assert isinstance(net, DistributedR3M)
        frames = torch.from_numpy(detgen.frames("ddp", (8, 3, 224, 224))).to("cuda:0")
        per = 8 // world
        g = _lp_backward(net, frames[rank * per:(rank + 1) * per])
        q.put((rank, g.cpu().numpy()))
        dist.barrier()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_two_rank_gradients_equal_single_process(hip, world):
    """world ranks (2, and 4 so that the slice / hook code sees ranks >= 2) share the test box's GPU over gloo, each with 8 / world
    frames: every rank ends with the same averaged gradient, equal to the single-process gradient of all 8 frames."""
    from oracle import detgen
    from r3m_amd.parallel import SingleDevice
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    for r, v in res.items():
        assert not isinstance(v, str), f"rank {r}: {v}"
    for r in range(1, world):
        np.testing.assert_array_equal(res[0], res[r])      # every rank holds the same averaged gradient
    single = SingleDevice(_build(18))
    frames = torch.from_numpy(detgen.frames("ddp", (8, 3, 224, 224))).to("cuda:0")
    g1 = _lp_backward(single, frames).cpu().numpy()
    err = np.abs(res[0] - g1).max() / np.abs(g1).max()
    print("ddp vs single max-rel", err)
    # not bit-equal: split-K / per-block partial sums group the 4-frame and 8-frame reductions differently (fp32 round-off)
    assert err < 2e-4


def _nccl_worker(port, q):
    """One rank, backend nccl (= RCCL): the production collective path — ReduceOp.AVG, async_op=True on RCCL's stream, joined
    by finish() — during a REAL staged backward with the language head, against the same step without any collective."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import detgen
        from r3m_amd import R3M
        from r3m_amd.parallel import DistributedR3M, SingleDevice, make_network_wrapper
        from r3m_amd.trainer import Trainer

        def build():
            m = R3M("cuda", 1e-4, 1024, size=18, l2weight=1e-5, l1weight=1e-5, langweight=1.0, tcnweight=1.0)
            shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
            m.convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict(shapes).items()})
            torch.manual_seed(11)
            m.lang_rew.reset_parameters()
            return m.to("cuda:0")

        B = 4
        frames = torch.from_numpy(detgen.frames("nccl1", (B, 5, 3, 224, 224))).to("cuda:0")
        feats = torch.from_numpy(detgen.uniform("langfeat", (B, 768), -0.6, 0.6)).to("cuda:0")
        out = {}
        for name in ("plain", "rccl"):
            m = build()
            assert isinstance(make_network_wrapper(m), SingleDevice)               # world 1 without force: no collectives
            net = SingleDevice(m) if name == "plain" else make_network_wrapper(m, force=True)
            if name == "rccl":
                assert isinstance(net, DistributedR3M) and net.sync.active and net.sync._avg
            steps = []
            for it in range(2):
                torch.manual_seed(5 + it)
                metrics, _ = Trainer(1).update(net, (frames, (feats, torch.ones(B))), it)
                torch.cuda.synchronize()
                steps.append((m.convnet.flat_grads().clone(), m.lang_rew.flat_grads().clone(), m.convnet.flat_params().clone(),
                              m.lang_rew.flat_params().clone(), metrics["full_loss"]))
            out[name] = steps
            if name == "rccl":
                # per step: the language-head buffer (first) + the encoder slices, all issued DURING backward; slices under 16 MB
                # wait for the next stage and go out merged (parallel.py): ResNet-18 sends layer4, then layer3 + layer2 + layer1/stem
                assert net.sync.launched == 2 * (1 + ENC_SLICES[18]), net.sync.launched
                assert net.sync._pending == []
        for it in range(2):
            for a, b, what in zip(out["plain"][it][:4], out["rccl"][it][:4], ("enc grads", "head grads", "enc params", "head params")):
                assert torch.equal(a, b), f"step {it}: {what} differ between the plain and the RCCL-synced run"
            assert out["plain"][it][4] == out["rccl"][it][4]
        assert not torch.equal(out["rccl"][0][2], out["rccl"][1][2])                # Adam consumed the synced gradients
        # a raw GradSync on a side stream: the collective must order after work queued on the CURRENT stream at enqueue
        from r3m_amd.parallel import GradSync
        s = GradSync(force=True)
        side = torch.cuda.Stream()
        buf = torch.zeros(1 << 24, device="cuda:0")
        with torch.cuda.stream(side):
            big = torch.randn(4096, 4096, device="cuda:0")
            for _ in range(8):
                big = big @ big * 1e-3                                              # keep the stream busy
            buf.fill_(3.0)
            s.reduce_slice(buf, 0, buf.numel())
            s.finish()
            chk = buf.sum()
        side.synchronize()
        assert float(chk) == 3.0 * buf.numel()
        q.put("ok")
    except Exception:  # noqa: BLE001
        import traceback
        q.put("FAIL: " + traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_one_rank_rccl_forced_sync_is_bit_identical(hip):
    """the `nccl` branch of GradSync (ReduceOp.AVG, async on RCCL's stream, finish()) executed for real."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=900)
    p.join(timeout=120)
    assert res == "ok", res


def test_bench_under_torchrun_one_rank(hip):
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the driver's N > 1 launch form with one rank —
    process-group init, RCCL all-reduces inside the step, barrier + max-over-ranks timing, one JSON line from rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--size", "18", "--clips-per-gpu", "8", "--langweight", "1", "--prewarm-seconds", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["collectives"].startswith(f"rccl all_reduce(AVG), {1 + ENC_SLICES[18]:.1f} per step")
    # the committed counter summary describes the ResNet-50 headline, not this ResNet-18 run: withheld, and the line says why
    assert out["roofline"]["traffic"] is None and "not this workload" in out["roofline"]["traffic_source"]


# ------------------------------------------------------------------------------------------------------------------------
# RCCL across devices (VERDICT r2 "next" #2). One process per GPU, backend "nccl" (= RCCL), world in {1, 2, 4, 8}: the
# world-1 case always runs (same worker code on the one-GPU test box: a one-rank group with forced collectives), the others
# arm themselves when that many GPUs are visible and are SKIPPED — not failed — otherwise.
# The reference's counterpart: nn.DataParallel at /root/reference/r3m/train_representation.py:27-31, r3m/__init__.py:72.
# ------------------------------------------------------------------------------------------------------------------------

def _need_gpus(world):
    have = torch.cuda.device_count()
    if have < world:
        pytest.skip(f"needs {world} GPUs, {have} visible")


def _rccl_init(rank, world, port):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    return dev


def _run_ranks(target, world, *extra, timeout=1500):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=timeout) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    for r, v in res.items():
        assert not (isinstance(v, str) and v.startswith("FAIL")), f"rank {r}: {v}"
    return res


def _guarded(fn, rank, world, port, q, *extra):
    try:
        q.put((rank, fn(rank, world, port, *extra)))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _equiv_rank(rank, world, port):
    """N ranks x B/N frames, BatchNorm on running statistics, LP loss -> the averaged gradient every rank holds
    after the staged RCCL all-reduce equals the one-process gradient on all B frames."""
    dev = _rccl_init(rank, world, port)
    from oracle import detgen
    from r3m_amd.parallel import DistributedR3M, make_network_wrapper
    m = _build(18, dev)
    net = make_network_wrapper(m, force=True)
    assert isinstance(net, DistributedR3M) and net.sync.active and net.sync._avg and net.sync.world == world
    frames = torch.from_numpy(detgen.frames("ddp", (8, 3, 224, 224))).to(dev)
    per = 8 // world
    core = net.module
    core.convnet.eval()
    core.encoder_opt.zero_grad()
    h = net(frames[rank * per:(rank + 1) * per])
    # the single-process loss is a mean over all 8 frames; the rank-local mean over 8/N frames, averaged over N ranks, is that
    loss = torch.linalg.norm(h, ord=2, dim=-1).mean() + 0.5 * torch.linalg.norm(h, ord=1, dim=-1).mean()
    loss.backward()
    assert net.sync.launched == ENC_SLICES[18]          # layer4, then layer3 + layer2 + layer1/stem merged, went out during backward
    net.finish_gradient_sync()
    torch.cuda.synchronize()
    return core.convnet.flat_grads().cpu().numpy()


def _equiv_worker(rank, world, port, q):      # module-level: mp "spawn" pickles the target by name
    _guarded(_equiv_rank, rank, world, port, q)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_rccl_ranks_reproduce_single_process_gradients(hip, world):
    from oracle import detgen
    from r3m_amd.parallel import SingleDevice
    _need_gpus(world)
    res = _run_ranks(_equiv_worker, world)
    for r in range(1, world):
        np.testing.assert_array_equal(res[0], res[r])      # RCCL AVG leaves the same bits on every rank
    single = SingleDevice(_build(18))
    frames = torch.from_numpy(detgen.frames("ddp", (8, 3, 224, 224))).to("cuda:0")
    g1 = _lp_backward(single, frames).cpu().numpy()
    err = np.abs(res[0] - g1).max() / np.abs(g1).max()
    print(f"rccl world {world} vs single process: max-rel {err:.3e}")
    if world == 1:
        np.testing.assert_array_equal(res[0], g1)          # mean over one rank is the identity, bit for bit
    assert err < 2e-4                                      # per-rank partial sums group the fp32 reductions differently


def _steps_rank(rank, world, port):
    """Three full Trainer.update steps (train-mode BatchNorm, LP + TCN + language loss through the reward head, Adam) on
    DIFFERENT clips per rank: parameters of the encoder and of the head must stay bit-identical across ranks."""
    dev = _rccl_init(rank, world, port)
    from oracle import detgen
    from r3m_amd import R3M
    from r3m_amd.parallel import make_network_wrapper
    from r3m_amd.trainer import Trainer
    torch.manual_seed(100 + rank)                          # ranks start from DIFFERENT weights: the constructor's broadcast must fix that
    m = R3M("cuda", 1e-3, 1024, size=18, l2weight=1e-5, l1weight=1e-5, langweight=1.0, tcnweight=1.0).to(dev)
    net = make_network_wrapper(m, force=True)
    B = 4
    frames = torch.from_numpy(detgen.frames(f"ddp3-{rank}", (B, 5, 3, 224, 224))).to(dev)
    feats = torch.from_numpy(detgen.uniform(f"ddp3f-{rank}", (B, 768), -0.6, 0.6)).to(dev)
    p0 = m.convnet.flat_params().clone()
    tr = Trainer(1)
    local = []
    for it in range(3):
        torch.manual_seed(7 + 13 * rank + it)              # shard-local negatives
        metrics, _ = tr.update(net, (frames, (feats, torch.ones(B))), it)
        local.append(metrics["full_loss"])
    torch.cuda.synchronize()
    out = {"loss": local, "launched": net.sync.launched}
    for name, p in (("enc", m.convnet.flat_params()), ("head", m.lang_rew.flat_params())):
        all_p = [torch.empty_like(p) for _ in range(world)]
        dist.all_gather(all_p, p.contiguous())
        out[name + "_identical"] = all(torch.equal(all_p[0], t) for t in all_p)
    out["moved"] = float((m.convnet.flat_params() - p0).abs().max())
    out["finite"] = bool(torch.isfinite(m.convnet.flat_params()).all())
    return out


def _steps_worker(rank, world, port, q):
    _guarded(_steps_rank, rank, world, port, q)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_rccl_parameters_stay_identical_over_three_steps(hip, world):
    _need_gpus(world)
    res = _run_ranks(_steps_worker, world)
    for r, o in res.items():
        assert o["enc_identical"] and o["head_identical"], (r, o)
        assert o["launched"] == 3 * (1 + ENC_SLICES[18]) and o["moved"] > 0 and o["finite"], (r, o)
    if world > 1:                                          # different clips per rank: the local losses must differ
        assert res[0]["loss"] != res[1]["loss"]


def _overlap_rank(rank, world, port, frames_per_rank):
    """Stream-ordering / overlap evidence on a ResNet-50 step: HIP events at the end of each backward stage (compute stream)
    and at the completion of each slice's all-reduce (an observer stream that joins RCCL's stream right after the enqueue)."""
    dev = _rccl_init(rank, world, port)
    from r3m_amd import R3M
    from r3m_amd.parallel import make_network_wrapper
    from r3m_amd.trainer import Trainer
    torch.manual_seed(1)
    B = frames_per_rank // 5
    m = R3M("cuda", 1e-4, 1024, size=50, l2weight=1e-5, l1weight=1e-5, langweight=0.0, tcnweight=1.0).to(dev)
    net = make_network_wrapper(m, force=True)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    frames = torch.randint(0, 256, (B, 5, 3, 224, 224), generator=g, device=dev, dtype=torch.int32).float()
    tr = Trainer(10 ** 9)
    for it in range(3):                                    # RCCL sets its channels up lazily: not in the measured step
        tr.update(net, (frames, [""] * B), it)
    torch.cuda.synchronize()
    obs = torch.cuda.Stream(device=dev)
    stage_ev, ar_ev = {}, {}
    conv = m.convnet
    inner = conv._stage_hook

    def stage_hook(stage, off, cnt):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()                                        # fires when this stage's last kernel retires
        stage_ev[stage] = ev
        inner(stage, off, cnt)

    def on_launch(idx, sl, work):
        with torch.cuda.stream(obs):
            work.wait()                                    # obs joins RCCL's stream for THIS collective
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(obs)
        ar_ev[idx] = (ev, sl.numel() * 4)

    conv._stage_hook = stage_hook
    base_idx = net.sync.launched
    net.sync.on_launch = on_launch
    net.sync.time_waits(True)
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    tr.update(net, (frames, [""] * B), 3)
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    torch.cuda.synchronize()
    exposed = net.sync.exposed_ms()
    net.sync.on_launch = None
    conv._stage_hook = inner
    t_stage = [base.elapsed_time(stage_ev[k]) for k in range(4)]
    n_ar = ENC_SLICES[50]                                  # 60 MB (stage 0), 28 MB (stage 1), 5 + 1 MB merged (sent when stage 3 ends)
    assert sorted(ar_ev) == [base_idx + k for k in range(n_ar)], sorted(ar_ev)
    t_ar = [base.elapsed_time(ar_ev[base_idx + k][0]) for k in range(n_ar)]
    return {"t_stage_end_ms": t_stage, "t_allreduce_done_ms": t_ar, "slice_bytes": [ar_ev[base_idx + k][1] for k in range(n_ar)],
            "step_ms": base.elapsed_time(end), "comm_exposed_ms": exposed}


def _overlap_worker(rank, world, port, q, frames_per_rank):
    _guarded(_overlap_rank, rank, world, port, q, frames_per_rank)


@pytest.mark.parametrize("world", [1, 2, 8])
def test_rccl_allreduce_overlaps_the_remaining_backward(hip, world):
    """The all-reduce of slice k (sent when backward stage k ends) must be complete before stage k+2 ends: it rides under the
    backward of the next stages instead of being exposed at the optimizer step."""
    _need_gpus(world)
    res = _run_ranks(_overlap_worker, world, 160)
    lines = []
    for r in sorted(res):
        o = res[r]
        lines.append(f"world {world} rank {r}: step {o['step_ms']:.2f} ms, comm exposed {o['comm_exposed_ms']:.3f} ms; stage ends "
                     + ", ".join(f"{t:.2f}" for t in o["t_stage_end_ms"]) + " ms; all-reduce done "
                     + ", ".join(f"{t:.2f}" for t in o["t_allreduce_done_ms"]) + " ms; slice MB "
                     + ", ".join(f"{b / 1e6:.1f}" for b in o["slice_bytes"]))
    print("\n".join(lines))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"ddp_overlap_world{world}.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    for r, o in res.items():
        ts, ta = o["t_stage_end_ms"], o["t_allreduce_done_ms"]
        assert all(ts[k] < ts[k + 1] for k in range(3)), (r, ts)
        sent_at = [0, 1, 3]                                # stage whose end sends slice k (the two small tail slices go as one)
        for k in range(3):
            assert ta[k] >= ts[sent_at[k]], (r, k, ts, ta) # a slice cannot be reduced before its stage produced it
        for k in range(2):
            assert ta[k] <= ts[k + 2], f"rank {r}: all-reduce of slice {k} finished at {ta[k]:.2f} ms, stage {k+2} ended at {ts[k+2]:.2f} ms"
        assert o["comm_exposed_ms"] < 0.25 * o["step_ms"], (r, o)


def _run_bench(extra, timeout=1500):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--size", "18", "--clips-per-gpu", "8",
           "--prewarm-seconds", "0", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks(hip):
    """`python bench.py --gpus N` with NO launcher in front (how the driver ran N = 1): N ranks over RCCL, one JSON line.
    N = min(2, visible GPUs); on a one-GPU box the launcher path is forced for N = 1."""
    n = min(2, torch.cuda.device_count())
    out = _run_bench(["--gpus", str(n)] + (["--force-launcher"] if n == 1 else []))
    assert out["n_gpus"] == n and out["rccl_ranks"] == n and out["value"] > 0
    assert out["config"]["collectives"].startswith(f"rccl all_reduce(AVG), {ENC_SLICES[18]:.1f} per step")
    # per-rank lines and the host placement every rank reports (NUMA node of its GPU from sysfs, its own cores, capped torch threads)
    assert len(out["ms_per_step_by_rank"]) == n and len(out["comm_exposed_ms_by_rank"]) == n
    assert max(out["ms_per_step_by_rank"]) == out["ms_per_step"]
    hb = out["host_binding"]
    assert len(hb) == n and all(b["local_rank"] == i for i, b in enumerate(hb))
    assert all(("error" in b) or (b["cpus"] and b["threads"] >= 1) for b in hb), hb
    assert out["config"]["frames_per_gpu"] == 40 and out["scaling"] == "weak"
    assert out["ms_per_step_rank_min"] <= out["ms_per_step"] == out["ms_per_step_rank_max"]
    assert out["comm_exposed_ms"] >= 0.0
    plain = _run_bench(["--gpus", "1"])                    # the one-process form is unchanged: no process group, no RCCL keys
    assert plain["n_gpus"] == 1 and "rccl_ranks" not in plain and plain["config"]["collectives"] == "none (single process)"


@pytest.mark.parametrize("world", [2, 8])
def test_bench_ranks_on_one_gpu_over_gloo(hip, world):
    """The N > 1 code of bench.py itself — self-spawned ranks, process group, barriers, all-reduced timing (max over ranks, per-rank
    min / max), gradient all-reduces inside the step — with TWO and with EIGHT ranks (the driver's largest launch: rank-indexed code
    sees ranks >= 2) on the one GPU of the test box (`--backend gloo --share-gpu`; RCCL refuses two ranks on a device, so the product
    backend is covered by the world-1 / armed tests above)."""
    out = _run_bench(["--gpus", str(world), "--backend", "gloo", "--share-gpu"])
    assert out["n_gpus"] == world and out["rccl_ranks"] == world and out["backend"] == "gloo" and out["value"] > 0
    assert out["config"]["collectives"].startswith(f"gloo all_reduce(SUM)/N, {ENC_SLICES[18]:.1f} per step")
    assert out["config"]["frames_per_gpu"] == 40 and out["config"]["parallelism"] == f"dp{world}"
    assert len(out["ms_per_step_by_rank"]) == world and len(out["comm_exposed_ms_by_rank"]) == world
    assert out["ms_per_step_rank_min"] <= out["ms_per_step_rank_max"] == out["ms_per_step"]
    # whole-job value = every rank's frames over the slowest rank's time
    assert abs(out["value"] - world * 40 * out["steps"] / (out["ms_per_step"] * out["steps"] * 1e-3)) <= 0.01 * out["value"]


# ------------------------------------------------------------------------------------------------------------------------
# global negatives (optional): the embeddings are all-gathered and the objective runs on the GLOBAL batch, as the
# reference's DataParallel step does on GPU 0 (/root/reference/r3m/trainer.py:41,87,136). Two ranks share the test box's one GPU
# over gloo (RCCL refuses two ranks on a device; the RCCL form of the same code arms itself below with >= 2 GPUs).
# ------------------------------------------------------------------------------------------------------------------------

def _gneg_rank(rank, world, port, backend):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        dev = _rccl_init(rank, world, port)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
    from oracle import detgen
    from r3m_amd import R3M
    from r3m_amd.parallel import DistributedR3M, SingleDevice, make_network_wrapper
    from r3m_amd.trainer import Trainer
    B = 4                                                  # global clips
    n = B // world
    frames = torch.from_numpy(detgen.frames("gneg", (B, 5, 3, 224, 224))).to(dev)
    feats = torch.from_numpy(detgen.uniform("gnegf", (B, 768), -0.6, 0.6)).to(dev)
    mask = torch.tensor([1.0, 0.0, 1.0, 1.0], device=dev)

    def build():
        torch.manual_seed(11)                              # same initial weights for the one-process and the two-rank model
        return R3M("cuda", 1e-3, 64, size=18, l2weight=1e-5, l1weight=1e-5, langweight=1.0, tcnweight=1.0).to(dev)

    out = {}
    # reference run in this process: the WHOLE batch, no wrapper collectives; BatchNorm on running statistics (eval=True) so that a
    # frame's embedding does not depend on which other frames share its forward
    single = SingleDevice(build())
    torch.manual_seed(77)
    m_single, _ = Trainer(1).update(single, (frames, (feats, mask)), 0, eval=True)
    net = make_network_wrapper(build(), global_negatives=True)
    assert isinstance(net, DistributedR3M) and net.global_negatives
    torch.manual_seed(77 if rank == 0 else 12345)          # rank 0's permutations are the ones used everywhere
    m_glob, _ = Trainer(1).update(net, (frames[rank * n:(rank + 1) * n], (feats[rank * n:(rank + 1) * n], mask[rank * n:(rank + 1) * n])),
                                  0, eval=True)
    out["eval_equal"] = m_glob == m_single
    out["m_glob"], out["m_single"] = m_glob, m_single
    # two training steps: every rank sees the same global objective -> identical metrics; parameters stay identical across ranks
    tr = Trainer(1)
    steps = []
    for it in range(2):
        torch.manual_seed(500 + 31 * rank + it)
        mm, _ = tr.update(net, (frames[rank * n:(rank + 1) * n], (feats[rank * n:(rank + 1) * n], mask[rank * n:(rank + 1) * n])), it)
        steps.append(mm)
    torch.cuda.synchronize()
    out["train_metrics"] = steps
    for name, p in (("enc", net.module.convnet.flat_params()), ("head", net.module.lang_rew.flat_params())):
        all_p = [torch.empty_like(p) for _ in range(world)]
        dist.all_gather(all_p, p.contiguous())
        out[name + "_identical"] = all(torch.equal(all_p[0], t) for t in all_p)
    out["finite"] = bool(torch.isfinite(net.module.convnet.flat_params()).all())
    return out


def _gneg_worker(rank, world, port, q, backend):
    _guarded(_gneg_rank, rank, world, port, q, backend)


def _check_gneg(res):
    for r, o in res.items():
        assert o["eval_equal"], (r, o["m_glob"], o["m_single"])   # 2 ranks x B/2 reproduce the 1-rank loss scalars bit for bit
        assert o["enc_identical"] and o["head_identical"] and o["finite"], (r, o)
    for r in res:
        assert res[0]["train_metrics"] == res[r]["train_metrics"]   # one global objective: the same numbers on every rank


@pytest.mark.parametrize("world", [2, 4])
def test_global_negatives_two_ranks_on_one_gpu(hip, world):
    """world = 4: one clip per rank — the row offsets of the gathered embeddings and each rank's gradient rows at ranks >= 2."""
    _check_gneg(_run_ranks(_gneg_worker, world, "gloo"))


def test_global_negatives_rccl(hip):
    _need_gpus(2)
    _check_gneg(_run_ranks(_gneg_worker, 2, "nccl"))


DEV = "cuda:0"


def test_replicated_inference_split_gather(hip):
    """`load_r3m(..., replicate=True)` -> parallel.ReplicatedInference: what the reference's DataParallel wrapper does for an inference
    batch (/root/reference/r3m/__init__.py:72) — split over the devices, one replica each, embeddings gathered on the first. One GPU
    here, so it is named three times: 7 frames go out as chunks of 3 + 3 + 1 through three replicas."""
    from r3m_amd import R3M
    from r3m_amd.parallel import ReplicatedInference, SingleDevice
    torch.manual_seed(3)
    m = R3M("cuda", 1e-4, 1024, size=18, langweight=0.0, tcnweight=1.0).to(DEV).eval()
    x = torch.randint(0, 256, (7, 3, 224, 224), device=DEV).float()
    rep = ReplicatedInference(m, devices=[DEV, DEV, DEV])
    with torch.no_grad():
        ref = SingleDevice(m)(x)
        out = rep(x)
    assert out.shape == ref.shape and out.device == ref.device
    torch.testing.assert_close(out, ref, rtol=2e-5, atol=1e-6)      # eval-mode frames are independent (chunk sizes pick other plans)
    assert len(rep._replicas) == 3 and rep._replicas[0] is m and rep._replicas[1] is not m
    assert list(rep.state_dict().keys()) == list(SingleDevice(m).state_dict().keys())      # replicas are not sub-modules
    with torch.no_grad():
        m.convnet.conv1.weight.mul_(0.5)                             # new weights: the replicas must follow
        ref2 = SingleDevice(m)(x)
        out2 = rep(x)
    assert not torch.allclose(ref2, ref)
    torch.testing.assert_close(out2, ref2, rtol=2e-5, atol=1e-6)
    # the framework's OWN updates write the flat buffers through raw pointers in native kernels — a FusedAdam step and the
    # running-statistics update of a train-mode forward bump no tensor version. The replicas must still follow.
    m.train()
    m.encoder_opt.zero_grad()
    m(x).sum().backward()
    m.encoder_opt.step()
    m.eval()
    with torch.no_grad():
        ref3 = SingleDevice(m)(x)
        out3 = rep(x)
    assert not torch.allclose(ref3, ref2)
    torch.testing.assert_close(out3, ref3, rtol=2e-5, atol=1e-6)
    assert torch.equal(rep._replicas[2].convnet.flat_params(), m.convnet.flat_params())
    assert torch.equal(rep._replicas[1].convnet._flat_b, m.convnet._flat_b)
    with pytest.raises(RuntimeError, match="forward-only"):
        rep(x)                                                       # grad mode: training goes through DistributedR3M
    with torch.no_grad():                                            # fewer frames than devices: empty chunks are skipped
        torch.testing.assert_close(rep(x[:2]), SingleDevice(m)(x[:2]), rtol=2e-5, atol=1e-6)
