#!/usr/bin/env python
"""bench.py — BASELINE.json metric: encoder frames/sec (fwd+bwd) ResNet-50 224^2 bs=256/GPU on MI355X.

    python bench.py --gpus 1 --steps K --warmup W                      # one process
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W    # one rank per GPU

A step = one R3M pre-training step on one batch of synthetic clips already resident in HBM: encoder forward over 5B
frames, LP + TCN loss, encoder backward (+ RCCL gradient all-reduce overlapped with it when N > 1), fused Adam — exactly
Trainer.update (BASELINE config 2: "ResNet-50 encoder fwd+bwd bs=256 fp32, TCN loss only, synthetic frames").
"bs=256" is 256 clips per GPU = 1280 frames per GPU (the reference's batch_size counts clips).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GFLOP_PER_FRAME = {50: 24.2868, 34: 21.7435, 18: 10.6453}   # algorithmic conv FLOPs fwd+bwd, SURVEY.md §8(d)
MB_PER_FRAME_BF16 = {50: 289.8, 34: 97.1, 18: 64.5}          # algorithmic HBM bytes fwd+bwd with 2-byte activations, SURVEY.md §8(d)
PEAK_HBM_GBS = 8000.0                                        # HBM3E spec (≈6300 GB/s achievable), MI355X_MICROARCH.md
PEAK_FP32_MFMA_TFLOPS = 157.3                                # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0                               # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16), not the 2:1-sparse figure
KCLASS = ["gather_gemm_128x128 (conv fwd/dgrad: gather, 3x3-window, 16-wide-K and persistent 1x1 kernels, all epilogues incl. the BatchNorm-backward partials)", "gather_gemm_256x64 (64-channel conv fwd/dgrad)", "wgrad_128x128",
          "wgrad_64x64"]


def cpu_baseline(size, clips, seconds_budget=25.0):
    """The oracle (PyTorch-CPU fp32 restatement of the reference step) timed on this box's host cores, bounded sample."""
    from oracle import r3m_ref
    torch.manual_seed(1)
    # physical cores of one socket are the useful ceiling for MKLDNN convs at this batch; 256 SMT threads thrash (measured
    # 0.22 frames/s at 256 threads on the 2 x EPYC 9575F host), so cap at 64 and report what was used
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_thr = max(1, min(64, avail // 2 if avail >= 4 else avail))
    torch.set_num_threads(n_thr)
    ref = r3m_ref.R3MRef(size=size, l2weight=1e-5, l1weight=1e-5, langweight=0.0, tcnweight=1.0)
    g = torch.Generator().manual_seed(1234)
    frames = torch.randint(0, 256, (clips, 5, 3, 224, 224), generator=g).float()
    perms = torch.stack([torch.randperm(clips) for _ in range(6)])
    t_w = time.time()
    r3m_ref.train_step_ref(ref, frames, tcn_perm=perms)   # warm-up
    t_w = time.time() - t_w
    times = []
    t_end = time.time() + seconds_budget
    min_steps = 3 if t_w < seconds_budget / 3 else 1
    while len(times) < min_steps or (time.time() < t_end and len(times) < 8):
        t0 = time.time()
        r3m_ref.train_step_ref(ref, frames, tcn_perm=perms)
        times.append(time.time() - t0)
        if time.time() > t_end and len(times) >= min_steps:
            break
    best = min(times)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(clips * 5 / best, 2), "unit": "frames/s", "cores": n_thr, "kind": "port",
            "sample": f"oracle/r3m_ref.py train step (ResNet-{size} fwd+loss+bwd+Adam, fp32, torch {torch.__version__} CPU), "
                      f"{clips} clips = {clips*5} frames, best of {len(times)} after 1 warm-up, {n_thr} threads on {cpu_model}"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # >= 20 timed steps after >= 5 warm-ups
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=50)
    ap.add_argument("--clips-per-gpu", type=int, default=256, help="clips per GPU (5 frames each); BASELINE bs=256")
    ap.add_argument("--langweight", type=float, default=0.0, help="> 0: BASELINE configs[2] (language head on frozen text features)")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                    help="bf16: BASELINE configs[2]/[4] (bf16 activations + bf16 MFMA, fp32 masters/statistics); default = headline fp32")
    ap.add_argument("--doaug", choices=["none", "rctraj", "rc"], default="none",
                    help="rctraj/rc: BASELINE configs[4] — every step starts from resident uint8 256x256 clips and runs the on-GPU "
                         "RandomResizedCrop(224) (csrc/augment.hip) inside the timed region")
    ap.add_argument("--unfused-crop", action="store_true", help="with --doaug: run the stand-alone crop kernel (fp32 frames written, then "
                                                                "read by the stem pre-pass) instead of cropping inside the stem pre-pass (A/B)")
    ap.add_argument("--encoder-only-frames", type=int, default=0,
                    help="> 0: the literal reading of 'bs=256': F frames through encoder forward + backward of sum|h| + Adam, no clip "
                         "structure / TCN loss (SURVEY.md §8(d) continuity point); the headline stays the 256-clip step")
    ap.add_argument("--prewarm-seconds", type=float, default=5.0,
                    help="untimed steps run BEFORE the --warmup steps until this much wall time has passed: an idle MI355X needs "
                         "seconds of sustained load to reach its clocks (measured: the same build reads 362-371 ms/step right after an "
                         "idle period with 2 warm-up steps and 348 ms with 12; 119.8 vs 105.6 ms in bf16). 0 disables.")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket the conv GEMM launches with HIP events (diagnostic: "
                                                                     "measures what the live roofline timing costs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=8)
    ap.add_argument("--launch-csv", default="", help="write one row per conv GEMM launch of the timed steps (layer report)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend; nccl (= RCCL) is the product path, gloo exists so that the N > 1 code of this script "
                         "(rank launch, barriers, max-over-ranks timing, secondary workloads on every rank) can run on a ONE-GPU box")
    ap.add_argument("--share-gpu", action="store_true",
                    help="with --backend gloo: rank r uses GPU r %% visible GPUs (all ranks on one device of a one-GPU box)")
    ap.add_argument("--force-launcher", action="store_true",
                    help="re-launch under torch.distributed.run also for --gpus 1 (the self-spawn path on a one-GPU box; N > 1 always does)")
    ap.add_argument("--no-bind", action="store_true",
                    help="N > 1: do not pin each rank to cores of its GPU's NUMA node (r3m_amd/utils/affinity.py)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short measurements of the other BASELINE configs that the default (headline) run appends under "
                         "'secondary' AFTER the headline's timed region")
    ap.add_argument("--secondary-steps", type=int, default=10)
    ap.add_argument("--secondary-warmup", type=int, default=5)
    return ap.parse_args(argv)


def is_headline(args):
    """The default workload = BASELINE configs[1]; only that run appends the 'secondary' measurements."""
    return (args.size == 50 and args.precision == "fp32" and args.langweight == 0 and args.doaug == "none"
            and args.encoder_only_frames == 0 and args.clips_per_gpu == 256)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_plan(args, environ, argv, port=None):
    """How `python bench.py --gpus N` becomes N ranks. Returns None when this process IS a rank (N == 1, or launched by
    torch.distributed.run: RANK/WORLD_SIZE in the environment), else the command line that re-runs this script under
    `python -m torch.distributed.run`, one rank per GPU over RCCL — the reference scales inside one process with
    nn.DataParallel (/root/reference/r3m/train_representation.py:27-31), so its users never type a launcher either.
    Pure function of its arguments (tests/test_bench_launch.py)."""
    if "RANK" in environ or "WORLD_SIZE" in environ:
        world = int(environ.get("WORLD_SIZE", "1"))
        if args.gpus != world:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                             f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
        return None
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus == 1 and not args.force_launcher:
        return None
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port if port is not None else free_port()),
            os.path.abspath(__file__)] + list(argv)


def gpu_count_error(gpus, world, local_rank, have, share_gpu):
    """Message for a rank that cannot get its own GPU (None when fine). Fails BEFORE the process group is created: a rank that
    dies inside init_process_group leaves the others waiting for the rendezvous timeout."""
    if share_gpu:
        return None if have >= 1 else "no GPU visible"
    if have < 1:
        return "no GPU visible to this process (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?)"
    if local_rank >= have:
        return (f"rank with LOCAL_RANK={local_rank} has no GPU: {have} GPU(s) visible but --gpus {gpus} / WORLD_SIZE={world} asks for one "
                f"per rank on this node — run with --gpus {have} or make more devices visible")
    return None


def self_spawn(cmd, n, share_gpu=False):
    have = torch.cuda.device_count()
    if have < n and not share_gpu:
        raise SystemExit(f"--gpus {n}: only {have} GPU(s) visible on this node")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    import subprocess
    return subprocess.call(cmd, env=env)


def workload_label(w, world):
    """Which BASELINE config a workload IS (never a label for work that is not executed)."""
    bf16 = w["precision"] == "bf16"
    if w["encoder_only_frames"] > 0:
        return "encoder-only continuity point"
    if bf16:
        lab = ("BASELINE configs[4]" if w["size"] == 34 and w["doaug"] == "rctraj" else
               "BASELINE configs[2]" if w["size"] == 50 and w["langweight"] > 0 else
               f"bf16 variant of the ResNet-{w['size']} step")
    elif w["size"] == 50 and w["langweight"] > 0:
        lab = "BASELINE configs[3] (full R3M loss: LP + TCN + language InfoNCE through the reward head)"
    elif w["size"] == 50:
        lab = "BASELINE configs[1]"
    else:
        lab = f"fp32 ResNet-{w['size']} variant of BASELINE configs[1]"
    if world > 1:
        lab += f", replicated on {world} GPUs (weak scaling, RCCL gradient mean overlapped with backward)"
    return lab


def measure(w, steps, warmup, prewarm_seconds, ctx, kernel_timing=True, launch_csv=""):
    """One workload `w` (dict: size, clips, precision, langweight, doaug, unfused_crop, encoder_only_frames): build the model,
    W warm-up steps, then EXACTLY K timed steps between barrier + synchronize; returns the JSON object (rank 0) or None."""
    import gc
    from r3m_amd import R3M, _lib
    from r3m_amd.parallel import make_network_wrapper
    from r3m_amd.trainer import Trainer
    L = _lib.lib()
    rank, world, dev, use_dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["use_dist"]
    size, B, precision = w["size"], w["clips"], w["precision"]

    torch.manual_seed(1)                               # config_rep.yaml seed
    model = R3M("cuda", 1e-4, 1024, size=size, l2weight=1e-5, l1weight=1e-5, langweight=w["langweight"], tcnweight=1.0,
                l2dist=True, bs=B, precision=precision)
    model = model.to(dev)
    net = make_network_wrapper(model, force=use_dist)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    if w["doaug"] == "none":
        frames = torch.randint(0, 256, (B, 5, 3, 224, 224), generator=g, device=dev, dtype=torch.int32).float()
        get_frames = lambda: frames
    else:
        from r3m_amd import augment
        raw = torch.randint(0, 256, (B, 5, 3, 256, 256), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
        box_gen = torch.Generator().manual_seed(99 + rank)
        get_frames = lambda: augment.random_resized_crop(raw, per_clip=(w["doaug"] == "rctraj"), generator=box_gen,
                                                         fused=not w["unfused_crop"])
    langs = [""] * B
    if w["langweight"] > 0:   # frozen DistilBERT stand-in: [B,768] N(0,1)*0.3, all clips have language
        gl = torch.Generator(device=dev).manual_seed(4321 + rank)
        langs = torch.randn((B, 768), generator=gl, device=dev) * 0.3
    trainer = Trainer(eval_freq=10 ** 9)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    enc_only = w["encoder_only_frames"] > 0
    if enc_only:
        Fe = w["encoder_only_frames"]
        xe = torch.randint(0, 256, (Fe, 3, 224, 224), generator=g, device=dev, dtype=torch.int32).float()

        class _EncTrainer:
            def update(self, net_, batch, step):
                core = net_.module
                core.train()
                h = net_(batch[0])
                loss = h.abs().sum()
                core.encoder_opt.zero_grad()
                loss.backward()
                sync = getattr(net_, "finish_gradient_sync", None)
                if sync is not None:
                    sync()
                core.encoder_opt.step()
                return {"full_loss": float(loss.item())}, ""

        trainer = _EncTrainer()
        get_frames = lambda: xe

    prewarm_steps = 0
    if prewarm_seconds > 0:                             # device warm-up (clock ramp), outside both the W warm-up and the timed K steps
        t_pw = time.perf_counter()
        while True:
            go = time.perf_counter() - t_pw < prewarm_seconds
            if use_dist:                                # every rank must run the same number of steps (collectives inside)
                flag = torch.tensor([1 if go else 0], device=dev, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                go = bool(flag.item())
            if not go:
                break
            trainer.update(net, (get_frames(), langs), 0)
            torch.cuda.synchronize()
            prewarm_steps += 1
    # Live kernel timing brackets conv GEMM launches with HIP events (two records per launch: a few microseconds of stream time each,
    # 2.8 ms per ResNet-50 step with all 167 launches bracketed). The roofline needs the DOMINANT class only: the LAST warm-up step
    # brackets all four classes (picks the dominant one and supplies the per-launch averages of the other three), the timed steps
    # bracket that class alone. With --launch-csv (evidence runs: per-shape report of every launch) the timed steps bracket all.
    warm_ms, warm_launches, warm_flops, warm_bytes = None, None, None, None
    for i in range(warmup):
        probe = kernel_timing and not launch_csv and i == warmup - 1
        if probe:
            torch.cuda.synchronize()
            L.r3m_profile_classes(0xF)
            L.r3m_profile_enable(1)
        trainer.update(net, (get_frames(), langs), i)
        if probe:
            torch.cuda.synchronize()
            warm_ms, warm_launches, warm_flops = (C.c_double * 4)(), (C.c_longlong * 4)(), (C.c_double * 4)()
            _lib.check(L.r3m_profile_collect(warm_ms, warm_launches, warm_flops), "profile_collect")
            warm_bytes = (C.c_double * 4)()
            _lib.check(L.r3m_profile_collect_bytes(warm_bytes), "profile_collect_bytes")
            L.r3m_profile_enable(0)
    timed_classes = 0xF
    if warm_ms is not None and max(warm_ms) > 0:
        timed_classes = 1 << max(range(4), key=lambda k: warm_ms[k])
    total_steps = prewarm_steps + warmup + steps
    sync = getattr(net, "sync", None)
    if sync is not None:
        sync.time_waits(True)                           # HIP events on the compute stream around the waits for RCCL
    L.r3m_profile_classes(timed_classes)
    L.r3m_profile_enable(1 if kernel_timing else 0)
    if launch_csv and rank == 0:
        _lib.check(L.r3m_profile_dump_to(launch_csv.encode()), "profile_dump_to")
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        metrics, _ = trainer.update(net, (get_frames(), langs), warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    ms, launches, flops = (C.c_double * 4)(), (C.c_longlong * 4)(), (C.c_double * 4)()
    _lib.check(L.r3m_profile_collect(ms, launches, flops), "profile_collect")
    L.r3m_profile_enable(0)
    L.r3m_profile_classes(0xF)
    L.r3m_profile_dump_to(None)
    comm_exposed_ms = sync.exposed_ms() / steps if sync is not None else None
    if sync is not None:
        sync.time_waits(False)
    dt_min = dt_max = dt
    by_rank = None
    if use_dist:
        mine = torch.tensor([dt, comm_exposed_ms or 0.0], device=dev, dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)                     # every rank's wall time and exposed communication (rank order)
        by_rank = [(float(t[0].item()), float(t[1].item())) for t in allr]
        dt_max, dt_min = max(b[0] for b in by_rank), min(b[0] for b in by_rank)
        comm_exposed_ms = max(b[1] for b in by_rank)
        dt = dt_max
    collectives = ((f"rccl all_reduce(AVG)" if dist.is_initialized() and dist.get_backend() == "nccl" else "gloo all_reduce(SUM)/N")
                   + f", {net.sync.launched / max(1, total_steps):.1f} per step"
                   + (" (one-rank group: identity, issued to exercise the path)" if world == 1 else "")) if use_dist \
        else "none (single process)"

    out = None
    if rank == 0:
        F_total = (w["encoder_only_frames"] if enc_only else 5 * B) * world
        bytes_k = (C.c_double * 4)()
        _lib.check(L.r3m_profile_collect_bytes(bytes_k), "profile_collect_bytes")
        fps = F_total * steps / dt
        kernels = []
        for k in range(4):
            if launches[k]:
                kernels.append({"kernel": KCLASS[k], "launches_per_step": launches[k] / steps,
                                "avg_launch_ms": ms[k] / launches[k], "ms_per_step": ms[k] / steps,
                                "tflops": flops[k] / (ms[k] * 1e-3) / 1e12,
                                "algorithmic_GBps": bytes_k[k] / (ms[k] * 1e-3) / 1e9,
                                "measured_over": f"the {steps} timed steps"})
            elif warm_ms is not None and warm_launches[k]:
                kernels.append({"kernel": KCLASS[k], "launches_per_step": float(warm_launches[k]),
                                "avg_launch_ms": warm_ms[k] / warm_launches[k], "ms_per_step": warm_ms[k],
                                "tflops": warm_flops[k] / (warm_ms[k] * 1e-3) / 1e12,
                                "algorithmic_GBps": warm_bytes[k] / (warm_ms[k] * 1e-3) / 1e9,
                                "measured_over": "the last warm-up step (not bracketed inside the timed steps)"})
        dom = max(range(4), key=lambda k: ms[k])
        ach = flops[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
        bf16 = precision == "bf16"
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
        alg_bytes_per_launch = bytes_k[dom] / max(1, launches[dom])
        traffic, traffic_source = pmc_traffic(bf16, size, B, KCLASS[dom])
        cfg_label = workload_label(w, world)
        out = {
            "metric": "encoder frames/sec (fwd+bwd) ResNet-50 224^2 bs=256/GPU" if size == 50 and B == 256 else
                      f"encoder frames/sec (fwd+bwd) ResNet-{size} 224^2 bs={B}/GPU",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "prewarm_steps": prewarm_steps,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": f"{cfg_label}: ResNet-{size} R3M step (encoder fwd + LP/TCN{'/language' if w['langweight'] > 0 else ''} loss + bwd + Adam), "
                                   f"{'bf16 activations / bf16 MFMA, fp32 master weights + statistics + Adam' if bf16 else 'fp32'}, "
                                   f"{B} clips = {5*B} frames of 224x224x3 per GPU, tcnweight=1 langweight={w['langweight']:g} l1=l2=1e-5 l2dist doaug={w['doaug']}",
                       "clips_per_gpu": B, "frames_per_gpu": 5 * B, "parallelism": f"dp{world}",
                       "collectives": collectives, "final_full_loss": metrics["full_loss"]},
            "roofline": {"bound": "mfma", "kernel": KCLASS[dom], "achieved": round(ach, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_ratio": round(traffic / alg_bytes_per_launch, 3) if traffic and alg_bytes_per_launch else None,
                         "avg_launch_ms": round(ms[dom] / max(1, launches[dom]), 4),
                         "algorithmic_gflop_per_launch": round(flops[dom] / max(1, launches[dom]) / 1e9, 3),
                         "algorithmic_mbytes_per_launch": round(alg_bytes_per_launch / 1e6, 2),
                         "whole_step_frac": round(fps / world * GFLOP_PER_FRAME[size] / 1e3 / peak, 4),
                         "kernels": kernels},
        }
        if use_dist:
            out["rccl_ranks"] = dist.get_world_size()                  # ranks of the process group (RCCL unless "backend" says gloo)
            out["backend"] = dist.get_backend()
            out["ms_per_step_rank_min"] = round(dt_min / steps * 1e3, 3)
            out["ms_per_step_rank_max"] = round(dt_max / steps * 1e3, 3)
            out["comm_exposed_ms"] = round(comm_exposed_ms, 4)     # per step, max over ranks: compute stream idle in work.wait()
            out["ms_per_step_by_rank"] = [round(b[0] / steps * 1e3, 3) for b in by_rank]
            out["comm_exposed_ms_by_rank"] = [round(b[1], 4) for b in by_rank]
            if ctx.get("binding") is not None:
                out["host_binding"] = ctx["binding"]               # per rank: NUMA node of its GPU, cores, torch threads
        if bf16:
            # with 2-byte activations every ResNet here is under the bf16 ridge -> the bounding roof is HBM.
            # achieved = algorithmic bytes of the dominant kernel class (operands read once + results written once, summed
            # per launch by the library) / its measured duration; the MFMA view stays available under "mfma".
            ach_bw = bytes_k[dom] / (ms[dom] * 1e-3) / 1e9 if ms[dom] > 0 else 0.0
            mf = out["roofline"]
            out["roofline"] = {"bound": "hbm", "kernel": KCLASS[dom], "achieved": round(ach_bw, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(ach_bw / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                               "traffic_ratio": mf["traffic_ratio"], "avg_launch_ms": mf["avg_launch_ms"],
                               "algorithmic_mbytes_per_launch": mf["algorithmic_mbytes_per_launch"],
                               "whole_step_frac": round(fps / world * MB_PER_FRAME_BF16[size] / 1e3 / PEAK_HBM_GBS, 4),
                               "mfma": {"achieved": mf["achieved"], "peak": mf["peak"], "unit": "TFLOP/s", "frac": mf["frac"],
                                        "whole_step_frac": mf["whole_step_frac"]},
                               "kernels": kernels}
        if enc_only:
            out["metric"] = f"encoder frames/sec (fwd+bwd) ResNet-{size} 224^2, {w['encoder_only_frames']} frames/GPU, encoder only (loss = sum|h|)"
            out["config"]["workload"] = (f"encoder-only continuity point (SURVEY.md §8(d)): ResNet-{size} forward + backward of sum|h| + Adam on "
                                         f"{w['encoder_only_frames']} frames of 224x224x3 per GPU, {'bf16 activations' if bf16 else 'fp32'}")
            out["config"]["frames_per_gpu"] = w["encoder_only_frames"]
    # release the activation arena (the dominant HBM allocation) before the next workload is built
    del trainer, net, model, get_frames
    gc.collect()
    torch.cuda.empty_cache()
    return out


def csrc_sha16():
    """sha256 over r3m_amd/csrc/*.{hip,h} (sorted by name), first 16 hex digits: identifies the kernel sources of THIS tree (the GPU
    box has no .git). tools/pmc_report.py stamps the PMC summaries with the same function."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "r3m_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "r3m_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(bf16, size, clips, dom_kernel):
    """HBM bytes per launch of the dominant class from the PMC passes (rocprofv3 --pmc cannot run inside this process): NOT
    measured in this run — read from the committed summary of the last counter run and labelled as such. The summary is only
    used when it describes THIS workload and THIS kernel class and its source file is still in the tree."""
    # one summary per (precision, encoder size): pmc_latest[_bf16][_r<size>].json, the un-suffixed names being ResNet-50's
    stem = "pmc_latest_bf16" if bf16 else "pmc_latest"
    pmc_name = next((n for n in (f"{stem}_r{size}.json", f"{stem}.json") if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
    if pmc_name is None:
        return None, None
    pmc_path = os.path.join(ROOT, "profiles", pmc_name)
    try:
        pj = json.load(open(pmc_path))
    except Exception as e:   # a broken summary must not take the bench line down with it
        return None, f"profiles/{pmc_name} unreadable: {e}"
    src = pj.get("source") or ""
    if not os.path.exists(os.path.join(ROOT, src)):
        return None, f"profiles/{pmc_name} names {src!r}, which is not in the tree: traffic withheld"
    want = pj.get("workload") or {}
    if (want.get("size"), want.get("clips")) != (size, clips):
        return None, f"profiles/{pmc_name} was collected on ResNet-{want.get('size')} / {want.get('clips')} clips: not this workload"
    tag = (pj.get("dominant_kernel") or "").split()[0:2]          # e.g. ["gather_gemm", "128x128"]
    if not tag or "_".join(tag) not in dom_kernel.replace(" ", "_"):
        return None, f"profiles/{pmc_name} describes {pj.get('dominant_kernel')!r}, the dominant class here is {dom_kernel.split(' (')[0]!r}"
    stamp = pj.get("csrc_sha16")
    here = csrc_sha16()
    build = ("collected on THIS build" if stamp == here else
             f"collected on ANOTHER build (csrc {stamp or 'unstamped: before round 5'}, this tree {here}): re-run tools/gpu_pmc.sh")
    return pj.get("dominant_kernel_hbm_bytes_per_launch"), (
        f"profiles/{pmc_name} <- {src} (separate rocprofv3 --pmc passes of an earlier run of this command, {build}; "
        f"FETCH_SIZE x2 + WRITE_SIZE, KiB units; not measured live)")


FWD_GFLOP_PER_FRAME = {18: 3.6286, 34: 7.3407, 50: 8.1743}     # 2 * MACs of the convolutions, 224 x 224 (a third of fwd + bwd)


def measure_inference(size, frames, precision, steps, warmup, ctx):
    """What every downstream user of R3M calls (/root/reference/r3m/__init__.py:72-75, r3m/example.py:19-33): `load_r3m(...).eval()`
    forward under no_grad — `frames` frames of 224 x 224 per call and per GPU, running-statistics BatchNorm, no backward, no optimizer.
    Same timing discipline as the training step (barrier + synchronize on both sides of exactly `steps` calls, max over ranks)."""
    from r3m_amd import R3M
    rank, world, dev, use_dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["use_dist"]
    torch.manual_seed(1)
    model = R3M("cuda", 1e-4, 1024, size=size, langweight=0.0, tcnweight=1.0, precision=precision).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(77 + rank)
    x = torch.randint(0, 256, (frames, 3, 224, 224), generator=g, device=dev, dtype=torch.int32).float()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(warmup):
            h = model(x)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            h = model(x)
        barrier()
        dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(h).all())
    del model
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    ms = dt / steps * 1e3
    tflops = FWD_GFLOP_PER_FRAME[size] * frames / ms                     # GFLOP / ms = TFLOP/s, per GPU
    peak = 2500.0 if precision == "bf16" else 157.3
    return {"metric": f"encoder frames/sec (forward only, eval) ResNet-{size} 224^2, {frames} frames/GPU", "value": round(frames * world / (dt / steps), 1),
            "unit": "frames/s", "ms_per_call": round(ms, 3), "steps": steps, "warmup": warmup, "dtype": "bf16" if precision == "bf16" else "f32",
            "n_gpus": world, "workload": f"load_r3m-style eval forward under no_grad: ResNet-{size}, {frames} frames of 224x224x3 per GPU, "
                                         f"{'bf16 activations' if precision == 'bf16' else 'fp32'}, running-statistics BatchNorm",
            "roofline": {"bound": "mfma", "achieved": round(tflops, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
                         "work": f"{FWD_GFLOP_PER_FRAME[size]} GFLOP per frame (forward convolutions) x {frames} frames per call, whole call"}}


SECONDARY = [   # the other single-node BASELINE configs, timed AFTER the headline so that it is unperturbed
    ("configs[2]", dict(size=50, clips=256, precision="bf16", langweight=1.0, doaug="none")),
    ("configs[3]", dict(size=50, clips=256, precision="fp32", langweight=1.0, doaug="none")),
    ("configs[4]", dict(size=34, clips=512, precision="bf16", langweight=0.0, doaug="rctraj")),
    # the literal reading of BASELINE.json's "bs=256/GPU": 256 FRAMES through the encoder alone, forward + backward + Adam (SURVEY.md §8(d)
    # continuity point; the headline reads it as 256 clips = 1280 frames, the batch `Trainer.update` sees at batch_size 256)
    ("encoder_only_256_frames", dict(size=50, clips=256, precision="fp32", langweight=0.0, doaug="none", encoder_only_frames=256)),
]


def main():
    args = parse_args()
    cmd = launch_plan(args, os.environ, sys.argv[1:])
    if cmd is not None:
        raise SystemExit(self_spawn(cmd, args.gpus, args.share_gpu))

    # dmabuf IPC: RCCL across processes needs it on this driver (exported on the GPU boxes; set here too so that a rank started by a
    # foreign launcher with a scrubbed environment still gets it — the HSA runtime reads it at the first HIP call, below)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # The GPU step needs ONE host thread (plus autograd's); torch's intra-op pool defaults to a thread per core (256 here), and
    # its wake-ups next to the launching thread cost up to +85 ms per 94 ms step of the launch-dense ResNet-34 bf16 workload
    # (measured round 3). cpu_baseline() sets its own thread count afterwards.
    torch.set_num_threads(4)
    if args.share_gpu and args.backend != "gloo":
        raise SystemExit("--share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
    have = torch.cuda.device_count()
    msg = gpu_count_error(args.gpus, world, local_rank, have, args.share_gpu)
    if msg:
        raise SystemExit(msg)
    dev_index = local_rank % max(1, have) if args.share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # Under torch.distributed.run (RANK/MASTER_PORT in the environment) the RCCL process group is ALWAYS created and the
    # gradient all-reduces are issued — also with one rank, where a mean over one rank is the identity: `torchrun
    # --nproc-per-node 1 bench.py --gpus 1` therefore executes the same collective / barrier / max-over-ranks code that N = 8 runs.
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    ctx = {"rank": rank, "world": world, "dev": dev, "use_dist": use_dist, "binding": None}
    affinity0 = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    if use_dist and not args.no_bind and not args.share_gpu:      # also a one-rank group under a launcher: same code as N = 8
        # one rank per GPU on a two-socket host: the launching thread (≈1000 launches per step) stays on cores of its GPU's NUMA
        # node, ranks get disjoint cores, torch's intra-op pool is capped (round 3: foreign threads next to the launcher cost up
        # to +85 ms on a 94 ms step)
        from r3m_amd.utils import affinity
        info = affinity.bind_rank(local_rank, local_world)
        infos = [None] * world
        dist.all_gather_object(infos, info)
        ctx["binding"] = infos

    w = dict(size=args.size, clips=args.clips_per_gpu, precision=args.precision, langweight=args.langweight, doaug=args.doaug,
             unfused_crop=args.unfused_crop, encoder_only_frames=args.encoder_only_frames)
    out = measure(w, args.steps, args.warmup, args.prewarm_seconds, ctx, kernel_timing=not args.no_kernel_timing,
                  launch_csv=args.launch_csv)
    if is_headline(args) and not args.no_secondary and args.secondary_steps > 0:
        sec = {}
        for name, sw in SECONDARY:
            sw = dict({"encoder_only_frames": 0}, **sw, unfused_crop=False)
            try:
                r = measure(sw, args.secondary_steps, args.secondary_warmup, min(args.prewarm_seconds, 2.0), ctx)
            except Exception as e:   # a secondary workload must never cost the headline its line
                if use_dist:
                    raise            # ranks would diverge: fail loudly
                r = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0:
                if "error" not in r:
                    r = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "n_gpus")} | {
                        "workload": r["config"]["workload"], "collectives": r["config"]["collectives"],
                        "comm_exposed_ms": r.get("comm_exposed_ms"),
                        "roofline": {k: v for k, v in r["roofline"].items() if k != "kernels"}}
                sec[name] = r
        # inference: the call every downstream user makes, fp32 and bf16
        for name, prec in (("inference_fp32", "fp32"), ("inference_bf16", "bf16")):
            try:
                r = measure_inference(50, 1280, prec, args.secondary_steps, args.secondary_warmup, ctx)
            except Exception as e:
                if use_dist:
                    raise
                r = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0:
                sec[name] = r
        if rank == 0:
            out["secondary"] = sec
        # (rounds 4-5 printed `kernel_timing_overhead_ms` from a second short leg here: a difference of two legs run a minute apart on a
        # power-limited part is good to +-1 ms — it read -1.85 ms in BENCH_r05 — so it is gone. What the brackets
        # cost was measured once with interleaved same-box legs: 0.3 ms per step, profiles/r05_event_fence_ab.txt.)
        if rank == 0 and not args.no_kernel_timing:
            out["kernel_timing_note"] = ("timed steps bracket the dominant kernel class only (fence-free HIP events, 99 launches): "
                                         "0.3 ms per step in the interleaved same-box A/B profiles/r05_event_fence_ab.txt")
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            if affinity0 is not None:
                os.sched_setaffinity(0, affinity0)       # the CPU baseline uses the host's cores, not the rank's NUMA slice
            out["cpu_baseline"] = cpu_baseline(args.size, args.cpu_clips)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
