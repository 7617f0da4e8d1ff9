"""Data parallelism for the R3M step: one process per GPU, RCCL over xGMI through torch.distributed (backend "nccl" is
RCCL on ROCm), gradient all-reduce overlapped with the remaining backward.

The reference uses single-process torch.nn.DataParallel (/root/reference/r3m/train_representation.py:27-31, r3m/__init__.py:72):
per-step replicate + scatter + gather + reduce-to-GPU-0, loss and Adam on GPU 0 only. Here every rank owns a full replica
and a shard of the clips; the only exchange is the gradient mean:

  * gradients already sit in ONE flat buffer in layer order, so a "bucket" is a slice — no flatten/unflatten copies;
  * the encoder backward is issued in 4 stages (layer4, layer3, layer2, layer1+stem); after each, the finished slice
    (60 MB, 28 MB, 5 MB, 1 MB for ResNet-50) goes out as an async all-reduce that runs while the next stage computes.
    xGMI is point-to-point (7 links x ~153 GB/s per GPU): few large messages keep every link busy, per-tensor ones don't.
    Slices under `min_slice_bytes` (16 MB) are held and merged with the next stage's (they are neighbours in the flat buffer:
    layer2's slice ends where layer1+stem's begins), so ResNet-50 sends 60 + 28 + 6 MB: the two latency-sized tail
    messages, which nothing but the last backward stage could hide, become one;
  * BatchNorm statistics stay per-rank (as the reference's DataParallel does per replica chunk); negatives are drawn within
    the rank's shard — or, with `global_negatives=True`, across the global batch as the reference does on
    GPU 0 (trainer.py:41,87,136): one all_gather of the [5B, D] embeddings, the objective evaluated on the gathered batch by
    every rank with permutations shared from rank 0, and each rank backpropagating its own rows of the gradient.

Both wrappers expose `.module` like DataParallel, which Trainer.update and the snapshot code rely on
(trainer.py:58-59,72,127,156-158; train_representation.py:126).
"""
import torch
import torch.distributed as dist
import torch.nn as nn


class GradSync:
    """Mean-all-reduce of slices of flat gradient buffers; device-agnostic (RCCL on GPUs, gloo on CPU in tests)."""

    def __init__(self, process_group=None, force=False):
        """force=True issues the collectives even on a one-rank group (mean over one rank = identity): the way the RCCL
        enqueue / stream-ordering path is exercised on a single-GPU box (tests/test_gpu_ddp.py, `torchrun --nproc-per-node 1
        bench.py`)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or force)
        self._pending = []   # (work, slice, needs_div)
        backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self._avg = backend == "nccl"   # RCCL reduces with AVG natively; gloo has SUM only
        self.launched = 0               # collectives enqueued so far (tests / bench report it)
        self._time_waits = False
        self._wait_events = []          # (before, after) HIP events on the compute stream around finish()'s waits
        self.on_launch = None           # test hook: callable(index, slice, work) right after a collective is enqueued

    def time_waits(self, on):
        """bench.py: bracket the waits of finish() with HIP events on the compute stream. The first event fires when the last
        backward kernel retires, the second once every outstanding all-reduce has been joined: their distance is the time the
        compute stream sat idle for communication (`comm_exposed_ms`); 0 when RCCL finished under the backward."""
        self._time_waits = bool(on)
        self._wait_events.clear()

    def exposed_ms(self):
        """Sum over the finish() calls since time_waits(True); synchronises on the recorded events."""
        tot = 0.0
        for a, b in self._wait_events:
            b.synchronize()
            tot += a.elapsed_time(b)
        self._wait_events.clear()
        return tot

    def reduce_slice(self, flat, offset, count):
        if not self.active or count == 0:
            return
        self.launched += 1
        sl = flat[offset:offset + count]
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        work = dist.all_reduce(sl, op=op, group=self.group, async_op=True)
        self._pending.append((work, sl, not self._avg))
        if self.on_launch is not None:
            self.on_launch(self.launched - 1, sl, work)

    def finish(self):
        timed = self._time_waits and self._pending and self._pending[0][1].is_cuda
        if timed:
            ev_a = torch.cuda.Event(enable_timing=True)
            ev_a.record()
        for work, sl, needs_div in self._pending:
            work.wait()           # on GPUs: the current stream waits for RCCL's stream; no host block
            if needs_div:
                sl.div_(self.world)
        if timed:
            ev_b = torch.cuda.Event(enable_timing=True)
            ev_b.record()
            self._wait_events.append((ev_a, ev_b))
        self._pending.clear()

    def broadcast(self, tensor, src=0):
        if self.active:
            dist.broadcast(tensor, src=src, group=self.group)


class _GatherRows(torch.autograd.Function):
    """y = concat over ranks of x (dim 0), differentiable. Every rank evaluates the SAME objective on y (same inputs, shared
    permutations), so d objective / d x_rank is rows [rank n, (rank + 1) n) of its own d objective / d y — no reduce-scatter. The
    parameter gradient of the global objective is the SUM over ranks of J_rank^T g_rank while the gradient sync AVERAGES, hence
    the factor `world`."""

    @staticmethod
    def forward(ctx, x, group, world, rank):
        x = x.contiguous()
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x, group=group)
        ctx.rank, ctx.world, ctx.n = rank, world, x.shape[0]
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n] * float(ctx.world), None, None, None


class SingleDevice(nn.Module):
    """world_size == 1 stand-in for the reference's DataParallel wrapper: only provides `.module`."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def finish_gradient_sync(self):
        pass


class ReplicatedInference(nn.Module):
    """Opt-in multi-GPU INFERENCE for `load_r3m` users: the reference returns `DataParallel(rep)`, which splits an inference batch
    over all visible GPUs and gathers the embeddings on GPU 0 (/root/reference/r3m/__init__.py:72). Here: one replica of the module
    per device (deep copies, rebuilt when the wrapped module's storage changes and re-filled with its current VALUES on every forward:
    three flat device-to-device copies for the encoder — optimizer steps and BatchNorm statistics updates happen in native kernels that
    no version counter sees), the batch split along dim 0 in
    device order, every chunk enqueued on its own device's stream (the chunks run concurrently), outputs concatenated on the first
    device. `.module` and the state-dict key set are those of SingleDevice. Forward only: training is DistributedR3M's job (one
    process per GPU) and a backward through this wrapper raises.
    `devices` may name one device several times (tests on a one-GPU box exercise the split / gather logic that way)."""

    def __init__(self, module, devices=None):
        super().__init__()
        self.module = module
        if devices is None:
            devices = [f"cuda:{i}" for i in range(torch.cuda.device_count())]
        self.devices = [torch.device(d) for d in devices]
        if not self.devices:
            raise RuntimeError("ReplicatedInference: no GPU visible")
        self._replicas = None      # plain list on purpose: replicas are not sub-modules (not in state_dict(), not moved by .to())
        self._stamp = None

    def _structure_stamp(self):
        """What a replica's STRUCTURE depends on: which tensors the module holds (a load_state_dict with assign, .to(), a re-flatten
        give new storage) and train / eval mode. The VALUES are not part of it: the framework's own updates — FusedAdam / FusedSGD
        steps, the train-mode BatchNorm running-statistics update — write the flat buffers through raw pointers in native kernels and
        never bump `tensor._version`, so values are re-broadcast on every forward instead (`_refresh`)."""
        return tuple(t.data_ptr() for t in list(self.module.parameters()) + list(self.module.buffers()))

    @staticmethod
    def _flat_sets(mod):
        """[(sub-module path, [flat tensors])] for every sub-module that keeps its state in flat buffers (HipResNet: parameters, BatchNorm
        buffers, batch counters; the language-reward head: parameters) + the ids of all tensors those buffers cover."""
        sets, covered = [], set()
        for name, m in mod.named_modules():
            fl = []
            if hasattr(m, "_flat_p") and callable(getattr(m, "flat_params", None)):
                fl.append(m.flat_params())
                for attr in ("_flat_b", "_flat_nbt"):
                    if isinstance(getattr(m, attr, None), torch.Tensor):
                        fl.append(getattr(m, attr))
                covered.update(id(t) for t in list(m.parameters(recurse=True)) + list(m.buffers(recurse=True)))
                sets.append((name, fl))
        return sets, covered

    def _refresh(self, rep, dev):
        """Copy the live module's values into a replica: one device-to-device copy per flat buffer (three for the encoder, one for the
        language head), tensor by tensor for whatever is not flat. Cheap next to a forward, and correct whatever wrote the values."""
        src_sets, covered = self._flat_sets(self.module)
        dst_sets, _ = self._flat_sets(rep)
        with torch.no_grad():
            for (n0, fs), (n1, fd) in zip(src_sets, dst_sets):
                assert n0 == n1 and len(fs) == len(fd)
                for a, b in zip(fs, fd):
                    b.copy_(a, non_blocking=True)
            for a, b in zip(list(self.module.parameters()) + list(self.module.buffers()), list(rep.parameters()) + list(rep.buffers())):
                if id(a) not in covered:
                    b.copy_(a, non_blocking=True)

    def _ensure_replicas(self):
        stamp = self._structure_stamp()
        first = next(self.module.parameters()).device
        if self._replicas is None or stamp != self._stamp:
            import copy
            self._replicas = []
            for i, d in enumerate(self.devices):
                if i == 0 and d == first:
                    self._replicas.append(self.module)
                else:
                    self._replicas.append(copy.deepcopy(self.module).to(d).train(self.module.training))
            self._stamp = stamp
            return
        src_stream_dev = first
        for rep, d in zip(self._replicas, self.devices):
            if rep is not self.module:
                import contextlib
                with (torch.cuda.device(src_stream_dev) if src_stream_dev.type == "cuda" else contextlib.nullcontext()):
                    self._refresh(rep, d)
        if first.type == "cuda" and any(d != first for d in self.devices):
            torch.cuda.current_stream(first).synchronize()     # the copies were enqueued on the source device's stream

    def finish_gradient_sync(self):
        pass

    def forward(self, x, *a, **k):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.module.parameters()):
            raise RuntimeError("ReplicatedInference is forward-only: wrap the call in torch.no_grad(); multi-GPU training is "
                               "DistributedR3M (one process per GPU, python -m torch.distributed.run)")
        self._ensure_replicas()
        chunks = [c for c in torch.chunk(x, len(self.devices), dim=0) if c.shape[0] > 0]
        outs = []
        import contextlib
        for rep, dev, c in zip(self._replicas, self.devices, chunks):
            rep.train(self.module.training)
            with (torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()):
                outs.append(rep(c.to(dev, non_blocking=True), *a, **k))
        first = self.devices[0]
        return torch.cat([o.to(first, non_blocking=True) for o in outs], 0)


class DistributedR3M(nn.Module):
    """One replica per rank. Construct AFTER torch.distributed.init_process_group and after moving `module` to its GPU."""

    LAST_STAGE = 3   # the engine issues backward in stages 0..3 (layer4, layer3, layer2, layer1 + stem)

    def __init__(self, module, process_group=None, force=False, min_slice_bytes=16 << 20, global_negatives=False):
        super().__init__()
        self.module = module
        self.sync = GradSync(process_group, force=force)
        # negatives across the GLOBAL batch (off by default: shard-local negatives need no exchange; SURVEY.md §8(e) "optional")
        self.global_negatives = bool(global_negatives)
        self._head_done = False
        self.min_slice_bytes = int(min_slice_bytes)
        self._held = None   # (offset, count) of finished slices not sent yet (smaller than min_slice_bytes so far)
        self._enc_sent = False   # the encoder's gradient slices of this step went out from the stage hooks
        self._stages_seen = 0    # stage hooks fired since the last finish_gradient_sync()
        # identical replicas: rank 0's parameters and BatchNorm buffers win
        for owner in self._owners():
            self.sync.broadcast(owner.flat_params())
        conv = module.convnet
        self.sync.broadcast(conv._flat_b)
        conv._stage_hook = self._on_stage

    def _owners(self):
        return list(self.module.encoder_opt.owners)

    def _reduce_heads(self):
        """Gradients of the owners other than the encoder (the language-reward head: 32.5 MB for ResNet-50). The head's
        backward is complete before the encoder's starts (the encoder needs d loss / d embeddings, which includes the head's
        input gradient), so its all-reduce goes out FIRST and rides under the whole encoder backward."""
        if self._head_done:
            return
        self._head_done = True
        for owner in self._owners():
            if owner is self.module.convnet:
                continue
            if getattr(owner, "has_grads", lambda: True)():
                g = owner.flat_grads()
                self.sync.reduce_slice(g, 0, g.numel())

    def _on_stage(self, stage, offset, count):
        if self._enc_sent and stage == 0:
            # a second encoder backward after the slices of this step went out: it would accumulate into buffers RCCL is reducing
            # (and gloo would divide twice). The encoder fires the hooks for the LAST live backward only (max_live_forwards > 1), so
            # this is a backward that came after one the encoder believed to be the last — e.g. two separate .backward() calls
            raise RuntimeError("r3m_amd.DistributedR3M: a second encoder backward ran before finish_gradient_sync(); sum the losses "
                               "and call backward() once per step (or call finish_gradient_sync() + the optimizer step in between)")
        if stage == 0:
            self._reduce_heads()
        self._stages_seen += 1
        if self._held is not None:
            ho, hc = self._held
            if offset + count == ho:          # backward walks the flat buffer downwards: the new slice ends where the held one begins
                count += hc
            elif ho + hc == offset:
                offset, count = ho, hc + count
            else:                             # not neighbours (never for the engine's stage order): send the held slice by itself
                self.sync.reduce_slice(self.module.convnet.flat_grads(), ho, hc)
            self._held = None
        if stage == self.LAST_STAGE:
            self._enc_sent = True
        if stage < self.LAST_STAGE and count * 4 < self.min_slice_bytes:
            self._held = (offset, count)
            return
        self.sync.reduce_slice(self.module.convnet.flat_grads(), offset, count)

    def _flush_held(self):
        if self._held is not None:
            self.sync.reduce_slice(self.module.convnet.flat_grads(), *self._held)
            self._held = None

    def forward(self, *a, **k):
        return self.module(*a, **k)

    # ---- global negatives (Trainer.update calls these when `global_negatives` is set) ----
    def gather(self, x):
        """[n, ...] per rank -> [world n, ...] in rank order on every rank; differentiable when x requires grad."""
        if not self.sync.active:
            return x
        rank = dist.get_rank(self.sync.group)
        if x.requires_grad and torch.is_grad_enabled():
            return _GatherRows.apply(x, self.sync.group, self.sync.world, rank)
        x = x.contiguous()
        parts = [torch.empty_like(x) for _ in range(self.sync.world)]
        dist.all_gather(parts, x, group=self.sync.group)
        return torch.cat(parts, 0)

    def share(self, t):
        """Rank 0's value of `t` on every rank (in place): the permutations of the global batch must be the same everywhere."""
        self.sync.broadcast(t, src=0)
        return t

    def check_replicas(self, raise_on_mismatch=True):
        """Are the replicas still identical? Every rank applies the same averaged gradients to the same parameters, so the flat
        parameter buffers must stay BIT-identical across ranks; nothing re-checks that after construction unless this is called
        (the training loop does at every snapshot). Compares an exact integer stamp (two wrapping int64 sums over the fp32 words) of each owner's parameters with
        two small all-reduces (MIN, MAX); non-finite values raise FloatingPointError instead of a divergence report; BatchNorm running statistics differ by design (per-rank statistics, as the reference's
        DataParallel normalises per replica chunk; snapshots carry rank 0's) and are only reported. Returns
        {"params_identical": bool, "param_spread": max |max - min| / (|max| + tiny), "bn_buffer_spread": same for the buffers}."""
        out = {"params_identical": True, "param_spread": 0.0, "bn_buffer_spread": 0.0}
        if not self.sync.active:
            return out

        def spread(t):
            """Bit-exact replica comparison without a float64 copy: the fp32 words are reinterpreted as int32 and stamped with two
            wrapping int64 sums (plain and position-weighted: a permutation or a compensating pair of changes moves the second).
            MIN / MAX all-reduce of the stamp: equal on every rank <=> the same bits everywhere (up to a 2^-64-ish collision).
            Non-finite values are detected first and reported as such — NaN != NaN would otherwise read as 'diverged'."""
            d = t.detach()
            finite = torch.isfinite(d).all().to(torch.int64).reshape(1)
            dist.all_reduce(finite, op=dist.ReduceOp.MIN, group=self.sync.group)
            if int(finite.item()) == 0:
                raise FloatingPointError("r3m_amd.DistributedR3M.check_replicas: non-finite parameter or buffer values on at least one rank "
                                         "(a loss blow-up, not a replica divergence)")
            w = d.contiguous().view(torch.int32).to(torch.int64)
            n = w.numel()
            CH = 1 << 22
            s1 = torch.zeros((), dtype=torch.int64, device=d.device)
            s2 = torch.zeros((), dtype=torch.int64, device=d.device)
            for o in range(0, n, CH):                                  # chunked: the int64 image of a chunk is 32 MB, not 8 bytes per weight
                c = w[o:o + CH]
                s1 += c.sum()
                s2 += (c * (torch.arange(o, o + c.numel(), device=d.device, dtype=torch.int64) % 65521 + 1)).sum()
            stamp = torch.stack([s1, s2])
            lo, hi = stamp.clone(), stamp.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.sync.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.sync.group)
            same = bool(torch.equal(lo, hi))
            if same:
                return 0.0, True
            # diverged: say by how much (float view of the values; costs a reduction, only on the failure path)
            v = d.double()
            fs = torch.stack([v.sum(), (v * v).sum()])
            flo, fhi = fs.clone(), fs.clone()
            dist.all_reduce(flo, op=dist.ReduceOp.MIN, group=self.sync.group)
            dist.all_reduce(fhi, op=dist.ReduceOp.MAX, group=self.sync.group)
            return float(((fhi - flo).abs() / (fhi.abs() + 1e-300)).max()), False

        for owner in self._owners():
            sp, same = spread(owner.flat_params())
            out["param_spread"] = max(out["param_spread"], sp)
            out["params_identical"] = out["params_identical"] and same
        out["bn_buffer_spread"] = spread(self.module.convnet._flat_b)[0]
        if raise_on_mismatch and not out["params_identical"]:
            raise RuntimeError(f"r3m_amd.DistributedR3M: parameter replicas have diverged across ranks (relative spread of the "
                               f"stamp {out['param_spread']:.3e}): a rank skipped a step, stepped outside finish_gradient_sync(), "
                               f"or loaded different weights after construction")
        return out

    def finish_gradient_sync(self):
        """Call after backward, before the optimizer step: waits for the slices launched during backward (and reduces the
        head gradients now if no encoder backward ran, e.g. a frozen encoder)."""
        self._reduce_heads()
        self._flush_held()          # a backward that stopped before the last stage (partial stage range) leaves nothing behind
        conv = self.module.convnet
        if self._stages_seen == 0 and conv._flat_g is not None and not conv._grad_fresh:
            # encoder gradients exist but no backward fired the stage hooks (several live forwards of which one was never
            # backpropagated): reduce the whole buffer now — correct, just not overlapped
            g = conv.flat_grads()
            self.sync.reduce_slice(g, 0, g.numel())
            if not getattr(self, "_warned_unoverlapped", False):
                self._warned_unoverlapped = True
                import warnings
                warnings.warn("r3m_amd.DistributedR3M: the encoder gradients of this step were all-reduced in one blocking piece after "
                              "backward (a live forward was never backpropagated, so no backward could start the overlapped slices); "
                              "results are correct, the comm / compute overlap is lost for such steps", RuntimeWarning)
        self.sync.finish()
        self._head_done = False
        self._enc_sent = False
        self._stages_seen = 0


def make_network_wrapper(model, force=False, global_negatives=False):
    """What `make_network` (train_representation.py:27-31) returns here: DistributedR3M when a process group exists and
    world_size > 1 (or `force`: one-rank group, collectives still issued), else the trivial wrapper."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        return DistributedR3M(model, force=force, global_negatives=global_negatives)
    return SingleDevice(model)
