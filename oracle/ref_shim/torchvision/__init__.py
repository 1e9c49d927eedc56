"""Minimal `torchvision` stand-in so the read-only reference tree (`/root/reference`) imports
in this container (no torchvision wheel, no network).  TEST INFRASTRUCTURE ONLY: used by
oracle/make_golden.py to run the reference itself as the live oracle.  Only
`ops.StochasticDepth` has behaviour (cvnets/layers/stochastic_depth.py:7 subclasses it);
the detection symbols are inert placeholders for import-time name resolution.
"""
from . import ops  # noqa: F401
__version__ = "0.0.shim"


def __getattr__(name):  # PEP 562: anything else the reference's data / detection stack names at import time is an inert placeholder
    if name.startswith("__"):
        raise AttributeError(name)
    from autostub import _make
    return _make(name)
