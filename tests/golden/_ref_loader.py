"""Import-time stub loader for the upstream reference (/root/reference).

TEST INFRASTRUCTURE, build-container only.  The GPU box has no /root/reference;
nothing under tests/ that runs there imports this module.  It is used by
`make_golden.py` (to produce the committed fixtures) and by the
`-m "not gpu"` oracle-pinning tests, which skip when the reference is absent.

The reference's Python needs third-party modules that are not installed here
(numba, kornia, cv2, torchvision, pytorch_lightning, wandb, ...).  We inject
permissive stand-in *modules* into sys.modules so that the reference's own
source imports and its render math (pure torch) executes verbatim.  No
reference source is copied: the stubs only satisfy `import` statements, with
two functional exceptions whose published semantics are restated here:
  * numba.jit(nopython=True)  -> identity decorator (functions then run as
    plain NumPy float64, which is numba's semantics for these scalar loops);
  * kornia.create_meshgrid(H, W, normalized_coordinates=False) -> (1,H,W,2)
    integer pixel grid, [...,0]=x=0..W-1, [...,1]=y=0..H-1 (kornia 0.6.1).
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("NEO360_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models"))


class _Anything:
    """Callable/attribute sink used for symbols the render path never touches."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        return _Anything()


def _stub(name, **attrs):
    m = _StubModule(name)
    m.__path__ = []  # behave like a package so submodule imports resolve
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def _create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
    assert not normalized_coordinates
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([gx, gy], dim=-1)[None]


def _identity_jit(*dargs, **dkw):
    if len(dargs) == 1 and callable(dargs[0]) and not dkw:
        return dargs[0]
    return lambda f: f


class _LightningModule(torch.nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass


_installed = False

# top-level packages that are absent here and only needed to satisfy `import`
_STUB_ROOTS = (
    "cv2", "wandb", "imageio", "piqa", "lpips", "dotmap", "open3d",
    "torch_efficient_distloss", "torchvision", "pytorch_lightning",
    "matplotlib", "PIL", "skimage", "numba", "kornia", "trimesh", "colorama",
)


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Resolve any (sub)module of an absent third-party root to a stub module."""

    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in _STUB_ROOTS and fullname not in sys.modules:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install_stubs():
    global _installed
    if _installed:
        return
    _installed = True
    # the finder sits LAST on sys.meta_path, so packages that are really
    # installed (matplotlib, PIL, ...) still resolve normally.
    sys.meta_path.append(_StubFinder())
    _stub("numba", jit=_identity_jit, njit=_identity_jit)
    _stub("kornia", create_meshgrid=_create_meshgrid)
    _stub("pytorch_lightning", LightningModule=_LightningModule)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load(modname):
    """Import a reference module (e.g. 'models.vanilla_nerf.model') under stubs."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    install_stubs()
    return importlib.import_module(modname)
