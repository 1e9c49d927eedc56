

def __getattr__(name):  # PEP 562: anything else the reference's data / detection stack names at import time is an inert placeholder
    if name.startswith("__"):
        raise AttributeError(name)
    from autostub import _make
    return _make(name)
