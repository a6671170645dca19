"""Import shim for the READ-ONLY reference at /root/reference (build container only).

Used only by tests/golden/make_golden.py to capture golden vectors from the reference's own
Python functions.  Nothing here is imported by the product, the GPU tests, smoke() or bench.py:
/root/reference does not exist on the GPU box.

Recipe: `.cuda()` -> identity, MagicMock modules for the third-party
imports that are absent here, and a bare `models` package object so sub-modules import without
executing models/__init__.py (which pulls torchvision).
"""
import os
import sys
import types
from unittest.mock import MagicMock

REF = "/root/reference"


def install():
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present; golden vectors can only be regenerated "
                           "in the build container")
    os.environ.setdefault("MPLBACKEND", "Agg")
    import torch

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    import importlib
    for name in ["spconv", "spconv.pytorch", "torchsparse", "torchsparse.nn",
                 "torchsparse.nn.functional", "torchsparse.nn.utils", "torchsparse.tensor",
                 "torchsparse.utils", "torchvision", "torchvision.utils", "trimesh", "skimage",
                 "skimage.measure", "loguru", "cv2", "pyrender", "pyvista", "transforms3d",
                 "numba", "yacs", "yacs.config", "tensorboardX", "memory_profiler"]:
        if name not in sys.modules:
            # the real module when it is installed (an environment that has torchsparse / spconv can then generate the
            # sparse-layer pin: make_golden.py gen_sparse_layers); a stand-in otherwise
            try:
                importlib.import_module(name)
                continue
            except Exception:  # noqa: BLE001  (ImportError, or a CUDA extension that fails to load)
                pass
            m = MagicMock()
            m.__all__ = []
            sys.modules[name] = m
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "models" not in sys.modules or not hasattr(sys.modules["models"], "__path__"):
        pkg = types.ModuleType("models")
        pkg.__path__ = [os.path.join(REF, "models")]
        sys.modules["models"] = pkg
    return torch


def have_real(*names):
    """True when every module in `names` is the real package, not the MagicMock stand-in install() leaves when it is absent"""
    return all(name in sys.modules and not isinstance(sys.modules[name], MagicMock) for name in names)
