"""not gpu: the N>1 path with world_size 2 on the gloo backend — gradient slices of a flat buffer are mean-reduced in the
order the encoder backward finishes its stages, parameters/buffers start identical on every rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from r3m_amd import R3M
        from r3m_amd.parallel import DistributedR3M, GradSync, make_network_wrapper
        torch.manual_seed(100 + rank)                      # different init per rank on purpose
        m = R3M("cpu", 1e-4, 1024, size=18, langweight=0.0, tcnweight=1.0)
        net = make_network_wrapper(m)
        assert isinstance(net, DistributedR3M) and net.module is m
        # rank 0's parameters / BN buffers were broadcast
        ref = m.convnet.flat_params().clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, m.convnet.flat_params())
        # emulate the encoder backward: fill the flat gradient buffer stage by stage and fire the stage hook
        g = m.convnet.flat_grads()
        n = g.numel()
        covered = 0
        for stage in range(4):
            off, cnt = m.convnet.stage_range(stage)
            g[off:off + cnt] = float(rank + 1) * torch.arange(off, off + cnt, dtype=torch.float32) / n
            m.convnet._stage_hook(stage, off, cnt)
            covered += cnt
        assert covered == n
        net.finish_gradient_sync()
        expect = (sum(range(1, world + 1)) / world) * torch.arange(n, dtype=torch.float32) / n
        torch.testing.assert_close(g, expect, rtol=1e-6, atol=1e-7)
        # param.grad views see the reduced values
        w = dict(m.convnet.named_parameters())["layer4.1.conv2.weight"]
        assert w.grad is not None and w.grad.data_ptr() >= g.data_ptr()
        # language head: its gradients go out FIRST (at the first stage hook), once per step, and finish() re-arms the step
        torch.manual_seed(200 + rank)
        m2 = R3M("cpu", 1e-4, 64, size=18, langweight=1.0, tcnweight=1.0)
        net2 = make_network_wrapper(m2)
        head = m2.lang_rew
        hg = head.flat_grads()
        for step in range(2):
            hg.fill_(float(rank + 1 + step))
            head._has_grads = True
            g2 = m2.convnet.flat_grads()
            g2.fill_(float(10 * (rank + 1)))
            before = net2.sync.launched
            for stage in range(4):
                off, cnt = m2.convnet.stage_range(stage)
                m2.convnet._stage_hook(stage, off, cnt)
                if stage == 0:
                    assert net2.sync.launched == before + 2          # head buffer + the layer4 slice
            assert net2.sync.launched == before + 5
            net2.finish_gradient_sync()
            assert net2.sync.launched == before + 5                  # nothing left for finish() to issue
            assert torch.allclose(hg, torch.full_like(hg, (sum(range(1, world + 1)) / world) + step))
            assert torch.allclose(g2, torch.full_like(g2, 10 * sum(range(1, world + 1)) / world))
        # a step without encoder backward (no stage hook fired): finish() still reduces the head
        hg.fill_(float(rank))
        head._has_grads = True
        before = net2.sync.launched
        net2.finish_gradient_sync()
        assert net2.sync.launched == before + 1 and torch.allclose(hg, torch.full_like(hg, (world - 1) / 2.0))
        # one-rank semantics of `force` are exercised on the GPU (tests/test_gpu_ddp.py); here: world 2 is active without it
        assert net2.sync.active
        # plain GradSync on an arbitrary buffer + no-op at count 0
        s = GradSync()
        buf = torch.full((10,), float(rank))
        s.reduce_slice(buf, 2, 5)
        s.reduce_slice(buf, 0, 0)
        s.finish()
        assert torch.allclose(buf[2:7], torch.full((5,), (world - 1) / 2.0)) and float(buf[0]) == float(rank)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gradient_sync_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, status in res:
        assert status == "ok", f"rank {rank}: {status}"
