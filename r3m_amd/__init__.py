"""r3m_amd — MI355X-native hot path of R3M representation pre-training behind the reference's Python surface.

    from r3m_amd import load_r3m, R3M          # same call signatures as `from r3m import load_r3m, R3M`

Mirrors /root/reference/r3m/__init__.py:15-75: VALID_ARGS, cleanup_config, remove_language_head, load_r3m.
Downloading checkpoints (gdown) is out of scope (no network): load_r3m uses ~/.r3m/r3m_XX/{model.pt,config.yaml} when
present and otherwise returns a freshly initialised encoder with the published architecture/config for that id.
"""
import copy
import os
import warnings
from os.path import expanduser

import torch

from .config import Cfg, instantiate, load_config
from .models_r3m import R3M
from .parallel import SingleDevice

VALID_ARGS = ["_target_", "device", "lr", "hidden_dim", "size", "l2weight", "l1weight", "langweight", "tcnweight", "l2dist", "bs"]

device = "cuda" if torch.cuda.is_available() else "cpu"

_MODEL_IDS = {"resnet50": ("r3m_50", 50), "resnet34": ("r3m_34", 34), "resnet18": ("r3m_18", 18)}


def cleanup_config(cfg):
    """Keep only R3M's constructor arguments, force the target/device, drop the language head (__init__.py:21-33)."""
    config = copy.deepcopy(cfg)
    agent = config["agent"]
    for key in list(agent.keys()):
        if key not in VALID_ARGS:
            del agent[key]
    agent["_target_"] = "r3m.R3M"
    config["device"] = device
    agent["device"] = device
    agent["langweight"] = 0   # downstream use is as a visual representation
    return agent


def remove_language_head(state_dict):
    for key in list(state_dict.keys()):
        if ("lang_enc" in key) or ("lang_rew" in key):
            del state_dict[key]
    return state_dict


def _default_config(size):
    here = os.path.dirname(os.path.abspath(__file__))
    return load_config(os.path.join(here, "cfgs", "config_rep.yaml"), [f"agent.size={size}", "agent.langweight=1.0"])


def load_r3m(modelid, replicate=False):
    """The reference's loader (/root/reference/r3m/__init__.py:44-75), local cache only. Returns a wrapper exposing `.module` like
    the reference's DataParallel: SingleDevice (the module on ONE GPU) by default; `replicate=True` returns
    parallel.ReplicatedInference, which splits an inference batch over all visible GPUs as DataParallel does."""
    if modelid not in _MODEL_IDS:
        raise NameError('Invalid Model ID')
    foldername, size = _MODEL_IDS[modelid]
    home = os.path.join(expanduser("~"), ".r3m")
    modelpath = os.path.join(home, foldername, "model.pt")
    configpath = os.path.join(home, foldername, "config.yaml")
    have_ckpt = os.path.exists(modelpath) and os.path.exists(configpath)
    modelcfg = load_config(configpath) if have_ckpt else _default_config(size)
    cleancfg = cleanup_config(modelcfg)
    rep = instantiate(cleancfg)
    if replicate:
        from .parallel import ReplicatedInference
        rep = ReplicatedInference(rep.to(device) if device != "cpu" else rep)
    else:
        rep = SingleDevice(rep)   # exposes `.module` and `module.`-prefixed state-dict keys like DataParallel
    if have_ckpt:
        sd = remove_language_head(torch.load(modelpath, map_location=torch.device(device))["r3m"])
        rep.load_state_dict(sd)
    else:
        warnings.warn(f"r3m_amd.load_r3m({modelid!r}): no checkpoint at {modelpath} and no network access in this build; "
                      f"returning randomly initialised weights", RuntimeWarning)
    return rep


def load_r3m_reproduce(modelid):
    """Checkpoints of the paper's ablations (__init__.py:77-113): only resolvable from a local ~/.r3m cache."""
    folders = {"r3m": "original_r3m", "r3m_noaug": "original_r3m_noaug", "r3m_nol1": "original_r3m_nol1",
               "r3m_nolang": "original_r3m_nolang"}
    if modelid not in folders:
        raise NameError('Invalid Model ID')
    home = os.path.join(expanduser("~"), ".r3m")
    modelpath = os.path.join(home, folders[modelid], "model.pt")
    configpath = os.path.join(home, folders[modelid], "config.yaml")
    if not (os.path.exists(modelpath) and os.path.exists(configpath)):
        raise FileNotFoundError(f"{modelpath} not found; downloading is not available in this build")
    cleancfg = cleanup_config(load_config(configpath))
    rep = SingleDevice(instantiate(cleancfg))
    rep.load_state_dict(remove_language_head(torch.load(modelpath, map_location=torch.device(device))["r3m"]))
    return rep


__all__ = ["R3M", "load_r3m", "load_r3m_reproduce", "cleanup_config", "remove_language_head", "VALID_ARGS", "Cfg"]
