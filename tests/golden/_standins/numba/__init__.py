"""Identity stand-in for numba (golden capture only)."""
import numpy as _np


class _Type:
    """Usable as a numpy dtype (`.dtype`), as `t[:, :]` and as a signature `t(...)`."""

    def __init__(self, dt):
        self.dtype = _np.dtype(dt)

    def __getitem__(self, item):
        return self

    def __call__(self, *a, **k):
        return self


float64 = _Type(_np.float64)
int32 = _Type(_np.int32)
int64 = _Type(_np.int64)


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not isinstance(args[0], _Type):
        return args[0]
    return lambda fn: fn


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not isinstance(args[0], _Type):
        return args[0]

    # signature form: the reference's kronecker_delta does `bool - bool`
    # (legal under numba, not for numpy bools) -> feed integer arrays as lists.
    def deco(fn):
        def wrapped(*a):
            a = [x.tolist() if isinstance(x, _np.ndarray) and x.dtype.kind in "iu" else x
                 for x in a]
            return fn(*a)
        return wrapped
    return deco
