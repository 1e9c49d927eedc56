from . import _Placeholder


class RoIAlign(_Placeholder):
    pass
