"""not gpu: the N>1 path with world sizes 2, 4 and 8 on the gloo backend — gradient slices of a flat buffer are mean-reduced in the
order the encoder backward finishes its stages, parameters/buffers start identical on every rank. Round 6: the
rank-indexed code (gather row offsets, global-negatives row ownership, slice merge, replica check, CPU-affinity slices,
ReplicatedInference chunking) had never seen a rank >= 2; every test here now runs at world 2, 4 and 8."""
import os
import socket

import pytest
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
        # replicas identical after construction; BatchNorm buffers too (broadcast). A rank that drifts is caught
        chk = net.check_replicas()
        assert chk["params_identical"] and chk["param_spread"] == 0.0 and chk["bn_buffer_spread"] == 0.0, chk
        keep = (m.convnet.flat_params()[5].clone(), m.convnet._flat_b[3].clone())
        if rank == 1:
            m.convnet.flat_params()[5] += 1.0
            m.convnet._flat_b[3] += 0.5                     # running statistics may differ (per-rank by design): reported, not an error
        chk = net.check_replicas(raise_on_mismatch=False)
        assert not chk["params_identical"] and chk["param_spread"] > 0 and chk["bn_buffer_spread"] > 0, chk
        try:
            net.check_replicas()
            raise AssertionError("diverged replicas were accepted")
        except RuntimeError as e:
            assert "diverged" in str(e)
        m.convnet.flat_params()[5] = keep[0]               # (the stamp is bit-sensitive: p + 1 - 1 would not pass)
        m.convnet._flat_b[3] = keep[1]
        assert net.check_replicas()["params_identical"]
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
        # slices under min_slice_bytes wait for the next stage and go out merged (ResNet-18: 33.6 MB, then 8.4 + 2.1 + 0.6 MB as one)
        sizes = [m2.convnet.stage_range(st)[1] * 4 for st in range(4)]
        n_enc, held = 0, 0
        for st in range(4):
            held += sizes[st]
            if st == 3 or held >= net2.min_slice_bytes:
                n_enc, held = n_enc + 1, 0
        assert n_enc == 2 and sizes[0] >= net2.min_slice_bytes > sizes[1] + sizes[2] + sizes[3]
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
                if stage in (1, 2):
                    assert net2.sync.launched == before + 2 and net2._held is not None   # small slices are held back
            assert net2.sync.launched == before + 1 + n_enc
            net2.finish_gradient_sync()
            assert net2.sync.launched == before + 1 + n_enc          # nothing left for finish() to issue
            assert torch.allclose(hg, torch.full_like(hg, (sum(range(1, world + 1)) / world) + step))
            assert torch.allclose(g2, torch.full_like(g2, 10 * sum(range(1, world + 1)) / world))
        # a step without encoder backward (no stage hook fired): finish() still reduces the head
        hg.fill_(float(rank))
        head._has_grads = True
        before = net2.sync.launched
        net2.finish_gradient_sync()
        assert net2.sync.launched == before + 1 and torch.allclose(hg, torch.full_like(hg, (world - 1) / 2.0))
        # a backward that ends early (stages 0..1 only) leaves a held slice: finish() sends it; min_slice_bytes=0 restores one
        # collective per stage
        g2 = m2.convnet.flat_grads()
        g2.fill_(float(rank + 1))
        hg.fill_(0.0)
        head._has_grads = True
        before = net2.sync.launched
        for stage in range(2):
            m2.convnet._stage_hook(stage, *m2.convnet.stage_range(stage))
        assert net2._held is not None
        net2.finish_gradient_sync()
        assert net2._held is None and net2.sync.launched == before + 3
        o1, c1 = m2.convnet.stage_range(1)
        assert torch.allclose(g2[o1:o1 + c1], torch.full((c1,), sum(range(1, world + 1)) / world))
        o2, c2 = m2.convnet.stage_range(2)
        assert torch.allclose(g2[o2:o2 + c2], torch.full((c2,), float(rank + 1)))       # stage 2 never ran: untouched
        net2.min_slice_bytes = 0
        head._has_grads = False
        before = net2.sync.launched
        for stage in range(4):
            m2.convnet._stage_hook(stage, *m2.convnet.stage_range(stage))
        net2.finish_gradient_sync()
        assert net2.sync.launched == before + 4
        # ---- several live forwards per step (max_live_forwards = 2, `(h1 + h2).backward()`): every backward accumulates into the
        # same flat buffer, so only the LAST outstanding one may start the asynchronous slice reductions. The native
        # engine is replaced by a stand-in that fills the buffer the way it does (overwrite, then accumulate) and honours fire_hooks.
        from r3m_amd.encoder import _EncoderFn
        torch.manual_seed(300 + rank)
        m3 = R3M("cpu", 1e-4, 64, size=18, langweight=0.0, tcnweight=1.0, max_live_forwards=2)
        net3 = make_network_wrapper(m3)
        conv = m3.convnet
        fired = []

        def fake_forward(x, training, crop=None, saved=False):
            from r3m_amd.encoder import _pick_slot
            slot, conv._ring_pos = _pick_slot([sl.live for sl in conv._ring], conv._ring_pos, saved)     # as the real forward does
            conv._last_forward = (slot, 0)
            return torch.ones((x.shape[0], conv.outdim))

        def fake_backward(dh, generation, si=0, fire_hooks=True):
            gb = conv.flat_grads()
            if conv._grad_fresh:
                gb.zero_()
            gb += float(rank + 1)
            conv._grad_fresh = False
            fired.append(fire_hooks)
            if fire_hooks and conv._stage_hook is not None:
                for stage in range(4):
                    conv._stage_hook(stage, *conv.stage_range(stage))

        conv._run_forward, conv._run_backward = fake_forward, fake_backward
        anchor = next(conv.parameters())
        x = torch.zeros((5, 3, 224, 224))
        for step in range(2):
            m3.encoder_opt.zero_grad()
            h1 = _EncoderFn.apply(x, anchor, conv, True, None)
            h2 = _EncoderFn.apply(x, anchor, conv, True, None)
            assert conv._awaiting == 2
            fired.clear()
            before = net3.sync.launched
            (h1.sum() + h2.sum()).backward()
            assert fired == [False, True] and conv._awaiting == 0, fired      # the hooks fired once, for the last backward
            net3.finish_gradient_sync()
            assert net3.sync.launched == before + 2                           # ResNet-18: two slices, each sent ONCE
            g3 = conv.flat_grads()
            assert torch.allclose(g3, torch.full_like(g3, 2.0 * sum(range(1, world + 1)) / world)), float(g3[0])
        # a forward whose graph is still alive when the other's backward runs: that backward may not fire the hooks, finish() reduces
        # the whole buffer itself (and says so once); when the graph is dropped the count falls back by itself
        import warnings
        m3.encoder_opt.zero_grad()
        h1 = _EncoderFn.apply(x, anchor, conv, True, None)
        h2 = _EncoderFn.apply(x, anchor, conv, True, None)
        fired.clear()
        before = net3.sync.launched
        h2.sum().backward()
        del h1
        assert fired == [False] and conv._awaiting == 0
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            net3.finish_gradient_sync()
        assert any("overlap is lost" in str(c.message) for c in caught)
        assert net3.sync.launched == before + 1 and conv._awaiting == 0
        # ... and the NEXT step overlaps again: a dropped forward leaves nothing behind
        m3.encoder_opt.zero_grad()
        hd = _EncoderFn.apply(x, anchor, conv, True, None)       # a grad-enabled forward nobody differentiates
        del hd
        h1 = _EncoderFn.apply(x, anchor, conv, True, None)
        fired.clear()
        h1.sum().backward()
        assert fired == [True]
        net3.finish_gradient_sync()
        m3.encoder_opt.zero_grad()
        g3 = conv.flat_grads()
        assert torch.allclose(g3, torch.full_like(g3, sum(range(1, world + 1)) / world))
        # two SEPARATE backward() calls in one step: the second would add into slices that are being reduced — refused
        m3.encoder_opt.zero_grad()
        h1 = _EncoderFn.apply(x, anchor, conv, True, None)
        h1.sum().backward()
        h2 = _EncoderFn.apply(x, anchor, conv, True, None)
        try:
            h2.sum().backward()
            raise AssertionError("second encoder backward before finish_gradient_sync() was accepted")
        except RuntimeError as e:
            assert "second encoder backward" in str(e)
        net3.finish_gradient_sync()
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


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gradient_sync_gloo(world):
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


def _gneg_worker(rank, world, port, q):
    """global_negatives (/root/reference/r3m/trainer.py:41,87,136 — DataParallel gathers the embeddings and GPU 0
    draws negatives from the WHOLE batch): `world` ranks x B/world clips, embeddings gathered, the oracle's objective evaluated on the gathered
    batch by every rank with rank 0's permutations, each rank backpropagating its own rows."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import r3m_ref
        from r3m_amd import R3M
        from r3m_amd.parallel import DistributedR3M
        B, D, Fin = (6 if world == 2 else 8), 32, 20      # global batch of 6 (8) clips; "encoder" = one linear map R^20 -> R^32
        g = torch.Generator().manual_seed(5)
        X = torch.randn((B, 5, Fin), generator=g)
        theta0 = torch.randn((Fin, D), generator=g) * 0.3
        feats = torch.randn((B, 768), generator=g) * 0.3
        mask = torch.tensor([1.0, 1.0, 0.0, 1.0, 1.0, 1.0, 0.0, 1.0][:B])
        tcn_perm = torch.stack([torch.randperm(B, generator=g) for _ in range(6)])
        lang_perm = torch.stack([torch.randperm(B, generator=g) for _ in range(9)])
        ref = r3m_ref.R3MRef(size=18, hidden_dim=16, l2weight=1e-2, l1weight=1e-2, langweight=1.0, tcnweight=1.0)
        ref.outdim = D
        torch.manual_seed(9)
        ref.lang_rew = r3m_ref.LanguageRewardRef(D, 16, 768)

        # single process, whole batch
        th = theta0.clone().requires_grad_(True)
        loss1, m1, _ = r3m_ref.r3m_loss_ref(ref, X @ th, tcn_perm, feats, mask, lang_perm)
        ref.zero_grad()
        loss1.backward()
        g_theta1 = th.grad.clone()
        g_head1 = torch.cat([p.grad.reshape(-1) for p in ref.lang_rew.parameters()])

        # two ranks, B/2 clips each, negatives across the global batch
        m = R3M("cpu", 1e-4, 16, size=18, langweight=0.0, tcnweight=1.0)
        net = DistributedR3M(m, global_negatives=True)
        assert net.global_negatives and not DistributedR3M(m).global_negatives      # off by default
        n = B // world
        th = theta0.clone().requires_grad_(True)
        local = X[rank * n:(rank + 1) * n] @ th
        alle = net.gather(local)
        assert alle.shape == (B, 5, D) and alle.requires_grad
        assert torch.equal(alle.detach(), X @ theta0)                               # rank order, bit-identical rows
        f_all, k_all = net.gather(feats[rank * n:(rank + 1) * n]), net.gather(mask[rank * n:(rank + 1) * n])
        assert torch.equal(f_all, feats) and torch.equal(k_all, mask) and not f_all.requires_grad
        # every rank draws its own permutations (different RNG streams); rank 0's win
        torch.manual_seed(1000 + rank)
        mine = torch.stack([torch.randperm(B) for _ in range(6)])
        shared = net.share(mine.clone())
        ref0 = mine.clone()
        dist.broadcast(ref0, src=0)
        assert torch.equal(shared, ref0)
        loss2, m2, _ = r3m_ref.r3m_loss_ref(ref, alle, tcn_perm, f_all, k_all, lang_perm)
        assert m2 == m1, (m1, m2)                                                   # the loss scalars of the 1-rank step, bit for bit
        ref.zero_grad()
        loss2.backward()
        # encoder: the gradient sync AVERAGES the ranks' parameter gradients -> the single-process gradient of the global objective
        gth = th.grad.clone()
        dist.all_reduce(gth)
        gth /= world
        torch.testing.assert_close(gth, g_theta1, rtol=1e-5, atol=1e-7)
        # head: evaluated on the global batch by every rank -> every rank already holds the full gradient (the mean changes nothing)
        g_head2 = torch.cat([p.grad.reshape(-1) for p in ref.lang_rew.parameters()])
        torch.testing.assert_close(g_head2, g_head1, rtol=1e-6, atol=1e-8)
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_global_negatives_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gneg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, status in res:
        assert status == "ok", f"rank {rank}: {status}"


@pytest.mark.parametrize("ndev", [2, 3, 5, 8])
def test_replicated_inference_chunking_with_a_fake_device_list(ndev):
    """parallel.ReplicatedInference (what `load_r3m(..., replicate=True)` returns on a multi-GPU host) on `ndev` stand-in devices — the
    CPU named `ndev` times: the batch splits into per-device chunks in device order, every replica but the first is a deep copy, the
    outputs come back concatenated in order, batches smaller than the device count skip the empty chunks, and a value change of the
    wrapped module that NO version counter sees (the framework's native kernels write parameters through raw pointers: ADVICE r5)
    reaches every replica before the next forward."""
    from r3m_amd.parallel import ReplicatedInference
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.BatchNorm1d(4)).eval()
    rep = ReplicatedInference(m, devices=["cpu"] * ndev)
    for n in (1, ndev - 1, ndev, ndev + 1, 3 * ndev + 2):
        x = torch.randn((n, 6))
        with torch.no_grad():
            out = rep(x)
            torch.testing.assert_close(out, m(x), rtol=1e-6, atol=1e-6)      # (a CPU GEMM rounds by batch size: not bit for bit)
    assert len(rep._replicas) == ndev and rep._replicas[0] is m and all(r is not m for r in rep._replicas[1:])
    # a raw-pointer write: same storage, same version counter, new values
    w = m[0].weight
    v0 = w._version
    import ctypes
    src_w, rm = torch.ones_like(w) * 0.25, m[1].running_mean
    src_rm = torch.ones_like(rm) * 2.0
    ctypes.memmove(w.data_ptr(), src_w.data_ptr(), w.numel() * 4)
    ctypes.memmove(rm.data_ptr(), src_rm.data_ptr(), rm.numel() * 4)
    assert w._version == v0 and float(w[0, 0]) == 0.25
    x = torch.randn((2 * ndev + 1, 6))
    with torch.no_grad():
        torch.testing.assert_close(rep(x), m(x), rtol=1e-6, atol=1e-6)
    for r in rep._replicas[1:]:
        assert torch.equal(r[0].weight, w) and torch.equal(r[1].running_mean, rm)
    with pytest.raises(RuntimeError, match="forward-only"):
        rep(x)
