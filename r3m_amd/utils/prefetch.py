"""Host -> HBM input pipeline: the next batch crosses PCIe on its own HIP stream while the current step computes.

The reference does `batch_f.cuda()` of fp32 frames inside the step loop (/root/reference/r3m/train_representation.py:104):
3.0 MB per clip, synchronous. Here frames travel as uint8 when the loader yields uint8 (4x fewer bytes), from pinned memory,
on a copy stream, and the compute stream only waits on the event of the batch it is about to use."""
import torch


class CudaPrefetcher:
    def __init__(self, loader_iter, device, transform=None):
        self.it = loader_iter
        self.device = device
        self.transform = transform          # optional GPU-side transform (e.g. the rc/rctraj crop), run on the copy stream
        self.stream = torch.cuda.Stream(device=device)
        self._next = None
        self.copy_events = None             # (start, end) timing events of the most recent host->HBM copy, on the copy stream
        self._preload()

    def _preload(self):
        try:
            frames, labels = next(self.it)
        except StopIteration:
            self._next = None
            return
        if not frames.is_pinned():
            frames = frames.pin_memory()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.stream):
            t0.record(self.stream)
            x = frames.to(self.device, non_blocking=True)       # uint8 stays uint8 across PCIe (4x fewer bytes than fp32)
            t1.record(self.stream)
            if self.transform is not None:
                x = self.transform(x)
            x = x.float()
        self.copy_events = (t0, t1)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self._next = (x, labels, ev)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        x, labels, ev = self._next
        torch.cuda.current_stream(self.device).wait_event(ev)
        x.record_stream(torch.cuda.current_stream(self.device))   # allocated on the copy stream, consumed on the compute stream
        self._preload()
        return x, labels
