"""Metric sink. The reference logs every scalar to wandb (/root/reference/r3m/utils/logger.py:124-158); observability is off
the compute path, so this is a JSONL writer with the same `log_metrics(metrics, step, ty)` call (wandb used only if present
and configured)."""
import json
import os


class Logger:
    def __init__(self, log_dir, use_tb=False, cfg=None):
        self._log_dir = str(log_dir)
        os.makedirs(self._log_dir, exist_ok=True)
        self._f = open(os.path.join(self._log_dir, "metrics.jsonl"), "a")
        self._wandb = None
        if cfg is not None and cfg.get("wandbproject"):
            try:
                import wandb
                wandb.init(project=cfg["wandbproject"], entity=cfg.get("wandbuser"), name=cfg.get("experiment"))
                self._wandb = wandb
            except Exception:  # noqa: BLE001
                self._wandb = None

    def log_metrics(self, metrics, step, ty):
        rec = {"step": int(step), "ty": ty}
        rec.update({k: float(v) for k, v in metrics.items()})
        self._f.write(json.dumps(rec) + "\n")
        self._f.flush()
        if self._wandb is not None:
            for k, v in metrics.items():
                self._wandb.log({f"{ty}/{k}": v}, step=step)
