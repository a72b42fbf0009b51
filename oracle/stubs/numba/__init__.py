"""Identity-decorator stand-in for third-party numba (absent here)."""


def jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f
