"""Import the REFERENCE's own hot-path files by path (authoring container only: /root/reference does not exist on the GPU
box) to generate golden vectors — tests/golden/make_golden.py is the only caller. Nothing of the reference is copied:
its modules are executed in place with empty stubs for packages the image lacks (hydra, torchvision, r3m.utils), and the
torchvision ResNet graph — un-vendored third-party code — is supplied by oracle/resnet_ref.py. SURVEY.md §8(c)."""
import importlib.util
import os
import sys
import types

REF = "/root/reference/r3m"


def available():
    return os.path.isdir(REF)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def load_reference():
    """Returns (models_r3m, models_language, trainer) modules of the reference, executed from /root/reference."""
    if _cache:
        return _cache["r3m"], _cache["lang"], _cache["trainer"]
    import numpy
    import numpy.core.numeric  # noqa: F401  (the reference does `from numpy.core.numeric import full`)
    from . import resnet_ref
    saved = {k: sys.modules.get(k) for k in ("r3m", "r3m.utils", "torchvision", "torchvision.utils", "torchvision.transforms",
                                              "torchvision.models", "hydra")}
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models", resnet18=resnet_ref.resnet18, resnet34=resnet_ref.resnet34, resnet50=resnet_ref.resnet50)
    tv.transforms = _stub("torchvision.transforms", Normalize=resnet_ref.Normalize)
    tv.utils = _stub("torchvision.utils", save_image=lambda *a, **k: None)
    _stub("hydra")
    pkg = _stub("r3m")
    pkg.utils = _stub("r3m.utils")
    try:
        lang = _load("_ref_models_language", os.path.join(REF, "models", "models_language.py"))
        r3m = _load("_ref_models_r3m", os.path.join(REF, "models", "models_r3m.py"))
        trainer = _load("_ref_trainer", os.path.join(REF, "trainer.py"))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _cache.update(r3m=r3m, lang=lang, trainer=trainer)
    return r3m, lang, trainer
