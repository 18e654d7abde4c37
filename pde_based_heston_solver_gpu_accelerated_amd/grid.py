"""Host-side grids: mirror of the reference's `Grid` (src/grid.hpp:4-23, src/grid.cpp:16-61) and of the
`std::vector<GridViews>` / `View<GridViews*>` batches its launchers take (src/grid_pod.hpp:8-111).
Grid generation itself is libhadi's hadi_make_grid (host code inside the library)."""
import ctypes as C

import numpy as np

from . import _native as nat

_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


class Grid:
    """Grid(m1, S, S_0, K, c, m2, V, V_0, d): sinh-stretched s- and v-grids with S_0 / V_0 inserted as
    nodes (the largest node is dropped).  Fields keep the reference's names."""

    def __init__(self, m1, S, S_0, K, c, m2, V, V_0, d):
        self.m1, self.m2 = int(m1), int(m2)
        self.Vec_s = np.empty(self.m1 + 1)
        self.Vec_v = np.empty(self.m2 + 1)
        self.Delta_s = np.empty(self.m1)
        self.Delta_v = np.empty(self.m2)
        rc = nat.lib().hadi_make_grid(self.m1, S, S_0, K, c, self.m2, V, V_0, d,
                                      _p(self.Vec_s), _p(self.Vec_v), _p(self.Delta_s), _p(self.Delta_v))
        if rc != nat.HADI_OK:
            raise nat.HadiError(rc, "hadi_make_grid: invalid arguments")

    def rebuild_variance_views(self, V_0_new, V=5.0, d=5.0 / 500):
        """GridViews::rebuild_variance_views (grid_pod.hpp:25-73)."""
        rc = nat.lib().hadi_rebuild_variance(self.m2, V_0_new, V, d, _p(self.Vec_v), _p(self.Delta_v))
        if rc != nat.HADI_OK:
            raise nat.HadiError(rc, "hadi_rebuild_variance: invalid arguments")

    def find_s_index(self, S_0):
        return nat.lib().hadi_find_s_index(self.m1, _p(self.Vec_s), S_0)

    def find_v0_index(self, V_0):
        """grid_pod.hpp:76-87 (returns 0 when V_0 is not a node)."""
        return nat.lib().hadi_find_v_index(self.m2, _p(self.Vec_v), V_0)


class GridViewsBatch:
    """n grids as the four [n][...] arrays the C ABI takes (the data of a `View<GridViews*>`)."""

    def __init__(self, grids):
        grids = list(grids)
        if not grids:
            raise ValueError("empty grid batch")
        self.m1, self.m2 = grids[0].m1, grids[0].m2
        if any(g.m1 != self.m1 or g.m2 != self.m2 for g in grids):
            raise ValueError("all grids of a batch must share (m1, m2)")
        self.n = len(grids)
        self.Vec_s = np.ascontiguousarray(np.stack([g.Vec_s for g in grids]))
        self.Vec_v = np.ascontiguousarray(np.stack([g.Vec_v for g in grids]))
        self.Delta_s = np.ascontiguousarray(np.stack([g.Delta_s for g in grids]))
        self.Delta_v = np.ascontiguousarray(np.stack([g.Delta_v for g in grids]))

    @classmethod
    def for_strikes(cls, m1, m2, S_0, V_0, strikes):
        """The grid every call site of the reference builds per strike:
        Grid(m1, 8*K, S_0, K, K/5, m2, 5.0, V_0, 5.0/500) (device_solver.cpp:677)."""
        return cls(Grid(m1, 8 * K, S_0, K, K / 5, m2, 5.0, V_0, 5.0 / 500) for K in strikes)

    def call_payoff(self, strikes):
        """U_0 = max(s - K, 0) on every v-row (device_solver.cpp:711-715); [n][m]."""
        k = np.asarray(strikes, dtype=np.float64).reshape(-1, 1)
        row = np.maximum(self.Vec_s - k, 0.0)
        return np.ascontiguousarray(np.tile(row, (1, self.m2 + 1)))

    def put_payoff(self, strikes):
        """U_0 = max(K - s, 0) on every v-row.  The time stepper is payoff-agnostic, but the boundary vectors it builds
        are the reference's call-type ones (hes_boundary_kernels.hpp:41-75: dU/ds = e^{-r_f tau} at s_max, U = s e^{-r_f tau}
        at v_max) -- the reference has no put path, so results with this payoff match the reference's ALGORITHM on the
        same input, not a put price."""
        k = np.asarray(strikes, dtype=np.float64).reshape(-1, 1)
        row = np.maximum(k - self.Vec_s, 0.0)
        return np.ascontiguousarray(np.tile(row, (1, self.m2 + 1)))

    def to(self, device):
        """Copies the four arrays to a torch device (bench / HBM-resident use)."""
        import torch
        out = object.__new__(GridViewsBatch)
        out.m1, out.m2, out.n = self.m1, self.m2, self.n
        for k in ("Vec_s", "Vec_v", "Delta_s", "Delta_v"):
            setattr(out, k, torch.from_numpy(getattr(self, k)).to(device))
        return out
