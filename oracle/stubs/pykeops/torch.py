"""Dense emulation of the pykeops.torch.LazyTensor subset that the reference loss uses
(reference call sites: src/losses/focus.py:129-137,159).  Third-party stand-in, test
infrastructure only.  K-min ties resolve to the lowest trajectory index (stable sort), which
is what a sequential K-min scan does; KeOps' own tie order is unpinned."""
import torch


class LazyTensor:
    def __init__(self, x, _dense=None):
        self.t = x if _dense is None else _dense

    def __sub__(self, o):
        return LazyTensor(None, self.t - o.t)

    def __pow__(self, p):
        return LazyTensor(None, self.t ** p)

    def abs(self):
        return LazyTensor(None, self.t.abs())

    def sum(self, dim):
        return LazyTensor(None, self.t.sum(dim))

    @property
    def shape(self):
        return tuple(self.t.shape)

    def _kmin(self, K, ax):
        d = self.t.detach()
        # stable ascending sort => ties keep the lower index first
        vals, idx = torch.sort(d, dim=ax, stable=True)
        sl = [slice(None)] * d.dim()
        sl[ax] = slice(0, K)
        return vals[tuple(sl)], idx[tuple(sl)]

    def argKmin(self, K, dim=None, axis=None):
        ax = dim if dim is not None else axis
        return self._kmin(K, ax)[1].movedim(ax, -1)

    def Kmin(self, K, dim=None, axis=None):
        ax = dim if dim is not None else axis
        return self._kmin(K, ax)[0].movedim(ax, -1)
