"""Pre-training driver — the loop of /root/reference/r3m/train_representation.py:33-153 (Workspace: seed, loaders, model,
train/eval cadence, snapshots with the reference's file layout) on the HIP path.

    python -m r3m_amd.train_representation batch_size=16 agent.size=50 dataset=synthetic train_steps=100
    python -m torch.distributed.run --nproc-per-node 8 -m r3m_amd.train_representation ...      # one rank per GPU (RCCL)

Differences that are deliberate: one process per GPU with gradient all-reduce instead of nn.DataParallel (r3m_amd/parallel.py),
`batch_size` is per GPU, uint8-valued frames cross PCIe as uint8 (4x less than the reference's fp32),
crops run on the GPU, metrics go to JSONL.
"""
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

from . import config as cfgmod
from .parallel import make_network_wrapper
from .trainer import Trainer
from .utils import utils
from .utils.data_loaders import R3MBuffer, SyntheticBuffer
from .utils.logger import Logger
from .utils.prefetch import CudaPrefetcher


def filter_frozen_text_keys(state_dict):
    """Reference snapshots written with langweight > 0 carry the frozen DistilBERT as `module.lang_enc.model.*`
    (it is a registered submodule there, /root/reference/r3m/models/models_language.py:19-20); here LangEncoder owns no
    parameters (frozen features are an input), so those keys are dropped on load — every other key stays under strict
    loading. The reverse direction (reference code loading a snapshot written here with langweight > 0) needs
    strict=False on the reference side for the same keys: INTEGRATION.md §3."""
    return {k: v for k, v in state_dict.items() if ".lang_enc." not in k and not k.startswith("lang_enc.")}


def make_network(cfg_agent, global_negatives=False):
    model = cfgmod.instantiate(cfg_agent)
    dev = torch.device("cuda", torch.cuda.current_device())
    model = model.to(dev)
    return make_network_wrapper(model, global_negatives=global_negatives)


class Workspace:
    def __init__(self, cfg, work_dir=None):
        self.work_dir = Path(work_dir or Path.cwd())
        self.cfg = cfg
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        # The training process's own torch CPU work is a few tiny tensors per step (permutations, crop boxes); a thread per core
        # of intra-op pool next to the kernel-launching thread only costs (measured: up to +85 ms per 94 ms ResNet-34 bf16 step).
        torch.set_num_threads(min(torch.get_num_threads(), 8))
        utils.set_seed_everywhere(cfg.seed + self.rank)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.logger = Logger(self.work_dir / f"logs_rank{self.rank}", use_tb=False, cfg=cfg if self.rank == 0 else None)
        if cfg.dataset == "ego4d":
            train_it = R3MBuffer(cfg.datapath, cfg.num_workers, "train", "train", alpha=cfg.alpha, datasources=["ego4d"], doaug=cfg.doaug)
            val_it = R3MBuffer(cfg.datapath, cfg.num_workers, "val", "validation", alpha=0, datasources=["ego4d"], doaug=0)
        elif cfg.dataset == "synthetic":
            train_it, val_it = SyntheticBuffer(cfg.seed + 17 * self.rank), SyntheticBuffer(cfg.seed + 1000003)
        else:
            raise NameError('Invalid Dataset')
        mk = lambda it: iter(torch.utils.data.DataLoader(it, batch_size=cfg.batch_size, num_workers=cfg.num_workers, pin_memory=True))  # noqa: E731
        # batches are copied to HBM (and cropped, for rc/rctraj) one step ahead on a copy stream
        self.train_loader = CudaPrefetcher(mk(train_it), self.device, self._gpu_transform(cfg.doaug, cfg.seed + 31 * self.rank))
        self.val_loader = CudaPrefetcher(mk(val_it), self.device, None)
        self.model = make_network(cfg.agent, bool(cfg.get("global_negatives", False)))
        self.timer = utils.Timer()
        self._global_step = 0
        if cfg.load_snap:
            self.load_snapshot(cfg.load_snap)

    @property
    def global_step(self):
        return self._global_step

    @staticmethod
    def _gpu_transform(doaug, seed=0):
        if doaug in ("rc", "rctraj"):
            from .augment import random_resized_crop
            # Boxes only: the resample itself happens inside the encoder's stem pre-pass (augment.CroppedClips). The boxes come from
            # their OWN generator: in the reference they are drawn in loader workers (data_loaders.py:88-102), never from the main
            # process's stream that Trainer.update's randperm draws consume (trainer.py:86-92,136-137).
            gen = torch.Generator().manual_seed(0x5EED ^ int(seed))
            return lambda x: random_resized_crop(x, per_clip=(doaug == "rctraj"), generator=gen, fused=True)
        return None

    def train(self):
        train_until_step = utils.Until(self.cfg.train_steps, 1)
        eval_every_step = utils.Every(self.cfg.eval_freq, 1)
        trainer = Trainer(self.cfg.eval_freq)
        while train_until_step(self.global_step):
            t0 = time.time()
            batch_f, batch_langs = next(self.train_loader)
            t1 = time.time()
            metrics, st = trainer.update(self.model, (batch_f, list(batch_langs)), self.global_step)
            t2 = time.time()
            self.logger.log_metrics(metrics, self.global_step, ty='train')
            if self.global_step % 10 == 0 and self.rank == 0:
                print(self.global_step, metrics)
                print(f'Sample time {t1-t0}, Update time {t2-t1}')
                print(st)
            if eval_every_step(self.global_step):
                with torch.no_grad():
                    batch_f, batch_langs = next(self.val_loader)
                    metrics, st = trainer.update(self.model, (batch_f, list(batch_langs)), self.global_step, eval=True)
                    self.logger.log_metrics(metrics, self.global_step, ty='eval')
                    if hasattr(self.model, "check_replicas"):
                        # every rank (two tiny collectives): the snapshot below is rank 0's — parameters must be the same bits on
                        # every rank; BatchNorm running statistics are per rank by design, the snapshot carries rank 0's
                        self.model.check_replicas()
                    if self.rank == 0:
                        print("EVAL", self.global_step, metrics)
                        self.save_snapshot()
            self._global_step += 1

    def save_snapshot(self):
        """Same files/keys as the reference (train_representation.py:123-130): {'r3m': state_dict [, 'global_step']} with
        `module.`-prefixed keys; optimizer state is added under a new key (the reference drops it)."""
        sdict = {"r3m": self.model.state_dict()}
        torch.save(sdict, self.work_dir / f'snapshot_{self.global_step}.pt')
        sdict["global_step"] = self._global_step
        sdict["encoder_opt"] = self.model.module.encoder_opt.state_dict()
        torch.save(sdict, self.work_dir / 'snapshot.pt')

    def load_snapshot(self, snapshot_path):
        payload = torch.load(snapshot_path, map_location="cpu")
        self.model.load_state_dict(filter_frozen_text_keys(payload['r3m']))
        if 'global_step' in payload:
            self._global_step = payload['global_step']
        else:
            print("No global step found")
        if 'encoder_opt' in payload:
            self.model.module.encoder_opt.load_state_dict(payload['encoder_opt'])


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    here = os.path.dirname(os.path.abspath(__file__))
    cfg = cfgmod.load_config(os.path.join(here, "cfgs", "config_rep.yaml"), argv)
    # dmabuf IPC between the ranks' HIP runtimes (RCCL fails with `hipIpcGetMemHandle: invalid argument` on this driver without
    # it); read by the HSA runtime at the first HIP call, i.e. set_device below
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    have = torch.cuda.device_count()
    if local_rank >= have:
        raise SystemExit(f"LOCAL_RANK={local_rank} but {have} GPU(s) visible: launch one rank per GPU (--nproc-per-node {max(have, 1)})")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if bool(cfg.get("bind_cpus", True)):
            # each rank's threads on cores of its GPU's NUMA node, disjoint from the other ranks' (utils/affinity.py). BEFORE the
            # process group exists (RCCL's proxy / watchdog threads inherit the mask; existing threads would keep the old one) and
            # before the loader workers fork. The rank takes its whole share of the node (launcher, autograd thread and
            # `num_workers` loader processes live there); on a host too small for that the mask is left alone.
            from .utils import affinity
            info = affinity.bind_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))), max_cpus=None,
                                      min_cpus=int(cfg.get("num_workers", 0)) + 2)
            print(f"[local rank {local_rank}] host binding: {info}", flush=True)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    root_dir = Path.cwd() / "r3moutput" / str(cfg.experiment)
    root_dir.mkdir(parents=True, exist_ok=True)
    ws = Workspace(cfg, root_dir)
    snapshot = root_dir / 'snapshot.pt'
    if snapshot.exists():
        print(f'resuming: {snapshot}')
        ws.load_snapshot(snapshot)
    ws.train()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
