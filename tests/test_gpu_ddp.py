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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(size):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import detgen
    from r3m_amd import R3M
    m = R3M("cuda", 1e-4, 1024, size=size, l2weight=1.0, l1weight=0.5, langweight=0.0, tcnweight=0.0)
    shapes = [(k, tuple(v.shape)) for k, v in m.convnet.state_dict().items()]
    m.convnet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in detgen.resnet_state_dict(shapes).items()})
    return m.to("cuda:0")


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
        g = _lp_backward(net, frames[rank * 4:(rank + 1) * 4])
        q.put((rank, g.cpu().numpy()))
        dist.barrier()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process(hip):
    from oracle import detgen
    from r3m_amd.parallel import SingleDevice
    world = 2
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
    np.testing.assert_array_equal(res[0], res[1])          # both ranks hold the same averaged gradient
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
                # per step: 4 encoder slices + 1 language-head buffer, all issued DURING backward (head first)
                assert net.sync.launched == 2 * 5, net.sync.launched
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
    assert out["config"]["collectives"].startswith("rccl all_reduce(AVG), 5.0 per step")
    assert out["roofline"]["traffic_source"] is None or "not measured live" in out["roofline"]["traffic_source"]
