"""TEST INFRASTRUCTURE ONLY.  Import-time stand-ins for third-party packages the reference's *data / metrics / engine* stack imports
but this container does not have (no network, no wheels): cv2, av, pycocotools, torchtext, ftfy, fvcore, coremltools, skimage, h5py,
pybase64 and every `torchvision.*` submodule the hand-written shim next to this file does not define.

`install()` appends a `sys.meta_path` finder that fabricates, for those top-level names only, modules whose attributes are inert
placeholder classes: subclassable (`class ImageNet(ImageFolder)`), callable, usable as decorators and as enum-like constants
(`cv2.INTER_LINEAR`).  Nothing here computes anything — code that actually CALLS into a fabricated symbol during a test is a test bug
and fails on the placeholder.  It exists so that `engine.training_engine`, `options.opts` and `data` of the read-only reference tree
import, and the reference's own `Trainer` can drive a model through the launcher (tests/test_launch_cpu.py).
"""
import abc
import importlib.abc
import importlib.machinery
import sys
import types

ABSENT = ("cv2", "av", "pycocotools", "torchtext", "ftfy", "fvcore", "coremltools", "skimage", "h5py", "pybase64", "torchvision",
          "pytorchvideo", "decord", "wandb", "boto3", "botocore", "torchaudio", "pyarrow_hotfix", "tensorboard", "tensorboardX", "bolt")


class _Meta(abc.ABCMeta):  # ABCMeta: the reference mixes these placeholders into ABC-derived dataset classes
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _make(f"{cls.__name__}.{name}")

    def __call__(cls, *a, **k):
        if cls.__dict__.get("_fabricated") and len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # used as a decorator
        return super().__call__(*a, **k)

    def __int__(cls):
        return 0

    def __index__(cls):
        return 0

    def __iter__(cls):
        return iter(())


def _make(name):
    return _Meta(name, (), {"_fabricated": True, "__init__": lambda self, *a, **k: None, "__module__": "autostub"})


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = _make(name)
        setattr(self, name, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        m.__version__ = "0.0.stub"
        return m

    def exec_module(self, module):
        pass


def install():
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.append(_Finder())  # appended: real packages and the hand-written torchvision shim win
