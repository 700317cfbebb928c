"""Alias package: `src.pipeline...`, `src.utils...`, `src.dataloader...` resolve to freepose_amd.src.* so code written
against the reference's import paths (e.g. `from src.pipeline.estimators.pose_estimator import DinoPoseEstimator`)
runs unchanged on this implementation."""
import importlib
import sys

_ALIASES = [
    "utils", "utils.bbox_utils", "pipeline", "pipeline.utils", "pipeline.retrieval", "pipeline.retrieval.dino",
    "pipeline.retrieval.renderer", "pipeline.estimators", "pipeline.estimators.pose_estimator",
    "pipeline.estimators.online_pose_estimator", "pipeline.estimators.tracking_refiner", "pipeline.estimators.scale_estimators",
    "pipeline.refiner_utils",
    "dataloader", "dataloader.template", "dataloader.bop",
]


class _LazyAlias:
    """meta-path finder mapping src.X -> freepose_amd.src.X on first import (keeps `import src` cheap)."""

    def find_spec(self, name, path=None, target=None):
        if name.startswith("src.") and name[4:] in _ALIASES:
            real = importlib.import_module("freepose_amd.src." + name[4:])
            sys.modules[name] = real
            return importlib.util.spec_from_loader(name, loader=_Loader(real))
        return None


class _Loader:
    def __init__(self, mod):
        self.mod = mod

    def create_module(self, spec):
        return self.mod

    def exec_module(self, module):
        pass


import importlib.util  # noqa: E402

sys.meta_path.insert(0, _LazyAlias())
