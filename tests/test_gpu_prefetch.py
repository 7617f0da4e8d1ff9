"""-m gpu: the host -> HBM input pipeline (r3m_amd/utils/prefetch.py; SURVEY.md §8(f)4 — the reference does a synchronous
`batch_f.cuda()` of fp32 frames inside the loop, /root/reference/r3m/train_representation.py:104)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _batches(n, shape, dtype=torch.uint8):
    g = torch.Generator().manual_seed(3)
    out = []
    for i in range(n):
        x = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
        out.append((x if dtype == torch.uint8 else x.float(), [f"clip {i}"] * shape[0]))
    return out


def test_uint8_batches_arrive_intact_and_in_order(hip):
    from r3m_amd.utils.prefetch import CudaPrefetcher
    data = _batches(4, (6, 5, 3, 32, 32))
    got = list(CudaPrefetcher(iter(data), DEV))
    assert len(got) == 4
    for (x, labels), (x0, labels0) in zip(got, data):
        assert x.device == DEV and x.dtype == torch.float32 and labels == labels0
        assert torch.equal(x.cpu(), x0.float())                      # uint8 values preserved exactly, converted on the GPU
    # fp32 loaders (the reference's format) pass through too
    data = _batches(2, (2, 5, 3, 16, 16), torch.float32)
    for (x, _), (x0, _) in zip(CudaPrefetcher(iter(data), DEV), data):
        assert torch.equal(x.cpu(), x0)
    assert list(CudaPrefetcher(iter([]), DEV)) == []


def test_gpu_transform_runs_on_the_copy_stream_and_is_ordered(hip):
    """transform (the rc / rctraj crop in training) runs behind the copy on the copy stream; the consumer's stream waits on the
    batch's event, so a consumer that immediately reads sees finished data even when the copy stream is slow."""
    from r3m_amd.utils.prefetch import CudaPrefetcher
    data = _batches(3, (4, 5, 3, 64, 64))
    seen_streams = []

    def transform(x):
        seen_streams.append(torch.cuda.current_stream(DEV).cuda_stream)
        big = torch.randn(2048, 2048, device=DEV)
        for _ in range(20):                                          # make the copy stream lag behind the host
            big = big @ big * 1e-3
        return x + (big[0, 0] * 0).to(x.dtype)                       # value-neutral dependency on the slow work

    pf = CudaPrefetcher(iter(data), DEV, transform)
    for (x, _), (x0, _) in zip(pf, data):
        assert torch.equal(x.cpu(), x0.float())
    assert len(seen_streams) == 3 and all(s == pf.stream.cuda_stream for s in seen_streams)
    assert pf.stream.cuda_stream != torch.cuda.current_stream(DEV).cuda_stream


def test_next_batch_copy_overlaps_compute(hip):
    """The copy of batch i+1 is issued when batch i is handed out and runs on its own stream WHILE the consumer's kernels run:
    its end event precedes the end of a long compute sequence that was enqueued before it (HIP events on both streams)."""
    from r3m_amd.utils.prefetch import CudaPrefetcher
    frames = [torch.randint(0, 256, (64, 5, 3, 224, 224), dtype=torch.uint8).pin_memory() for _ in range(3)]   # 48 MB each (uint8)
    pf = CudaPrefetcher(iter([(f, ["x"] * 64) for f in frames]), DEV)
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream(DEV)
    base, c_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a = torch.randn(8192, 8192, device=DEV)
    base.record(cur)
    for _ in range(12):                                              # "the step": ~100+ ms of matrix work on the compute stream
        a = (a @ a) * 1e-4
    c_end.record(cur)
    x, _ = next(pf)                                                  # hands out batch 0 and ISSUES the copy of batch 1 right now
    t0, t1 = pf.copy_events
    torch.cuda.synchronize()
    compute_ms = base.elapsed_time(c_end)
    copy_start, copy_end = base.elapsed_time(t0), base.elapsed_time(t1)
    print(f"compute 0..{compute_ms:.1f} ms, copy of the next batch {copy_start:.1f}..{copy_end:.1f} ms "
          f"({frames[1].numel() / 1e6 / max(copy_end - copy_start, 1e-3):.1f} GB/s)")
    assert compute_ms > 20.0
    assert copy_end < compute_ms, "the H2D copy waited for the compute stream instead of overlapping it"
    assert torch.equal(x.cpu(), frames[0].float())


def test_small_uploads_survive_a_host_that_runs_far_ahead(hip):
    """_lib.upload_small stages per-step tensors (crop boxes, permutations) in a ring of pinned buffers. A caller that never
    synchronises (an encoder-only loop, a deep prefetcher) can be more than a ring ahead of the GPU: a slot whose copy has not
    run yet must not be overwritten (silently corrupted boxes). 64 uploads are queued behind ~0.3 s of GPU work."""
    from r3m_amd import _lib
    _lib._pinned.clear()
    a = torch.randn((8192, 8192), device=DEV)
    torch.cuda.synchronize()
    for _ in range(24):
        a = (a @ a) * 1e-2                                           # keeps the stream busy while the host queues the uploads
    outs = [_lib.upload_small(torch.full((4, 6), i, dtype=torch.int64), DEV, torch.int32) for i in range(64)]
    ring = _lib._pinned[((4, 6), torch.int32)][0]
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert o.dtype == torch.int32 and o.device == DEV
        assert torch.equal(o.cpu(), torch.full((4, 6), i, dtype=torch.int32)), f"upload {i} was overwritten before its copy ran"
    assert len(ring) >= 4                                            # the ring grew instead of recycling busy slots ...
    n = len(ring)
    for i in range(16):                                              # ... and with the GPU idle it stops growing
        _lib.upload_small(torch.full((4, 6), i, dtype=torch.int64), DEV, torch.int32)
        torch.cuda.synchronize()
    assert len(ring) == n
