"""Host-side mirror of the reference's batched launchers over libhadi's C ABI.

Names, argument order and meaning follow the reference (`parallel_DO_solve`,
src/device_solver.hpp:52-79; `compute_base_prices*` / `compute_jacobian*`,
src/jacobian_computation.hpp:43-231); the Kokkos struct arrays that only carry scratch
(A0/A1/A2 solvers, bounds_d) have no counterpart because the library owns its scratch.

Arrays may be numpy float64 (host memory, staged by the library) or torch CUDA tensors
(HBM-resident, passed as device pointers).  All arrays of one call must live in the same space.
"""
import ctypes as C

import numpy as np

from . import _native as nat
from ._native import EU, AM, DIV, AM_DIV, CALL, PUT, HadiError  # noqa: F401  (re-exported)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _is_device(x):
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


def _check_array(x, name, count):
    if _is_device(x):
        import torch
        if x.dtype != torch.float64 or not x.is_contiguous():
            raise ValueError("%s must be a contiguous float64 tensor" % name)
        if x.numel() != count:
            raise ValueError("%s has %d elements, expected %d" % (name, x.numel(), count))
        return C.cast(C.c_void_p(x.data_ptr()), _dp), True
    if not isinstance(x, np.ndarray) or x.dtype != np.float64 or not x.flags.c_contiguous:
        raise ValueError("%s must be a C-contiguous float64 numpy array or CUDA tensor" % name)
    if x.size != count:
        raise ValueError("%s has %d elements, expected %d" % (name, x.size, count))
    return x.ctypes.data_as(_dp), False


def _host_f64(x):
    return None if x is None else np.ascontiguousarray(np.asarray(x, dtype=np.float64))


class Dividends:
    """The three dividend views of the reference (device_solver.hpp:409-413)."""

    def __init__(self, dates, amounts, percentages):
        self.dates, self.amounts, self.percentages = _host_f64(dates), _host_f64(amounts), _host_f64(percentages)
        if not (len(self.dates) == len(self.amounts) == len(self.percentages)):
            raise ValueError("dividend arrays differ in length")

    def __len__(self):
        return len(self.dates)


class DOWorkspace:
    """DO_Workspace<Device> (src/DO_solver_workspace.hpp:4-44).  Only `U` carries data across the
    boundary (initial condition in, solution out); the reference's other eleven arrays are scratch
    and live inside the library handle here.  `lambda_bar` is filled by the American variants."""

    def __init__(self, nInstances, total_size, device=None):
        if device is None:
            self.U = np.zeros((nInstances, total_size))
            self.lambda_bar = np.zeros((nInstances, total_size))
        else:
            import torch
            self.U = torch.zeros((nInstances, total_size), dtype=torch.float64, device=device)
            self.lambda_bar = torch.zeros((nInstances, total_size), dtype=torch.float64, device=device)


class HestonADI:
    """One library handle = one GPU + one HIP stream (the reference's single Kokkos device)."""

    def __init__(self, device_id=0, lib_path=None):
        self._lib = nat.lib(lib_path)
        h = C.c_void_p()
        rc = self._lib.hadi_create(C.byref(h), int(device_id))
        if rc != nat.HADI_OK:
            raise HadiError(rc, self._lib.hadi_status_string(rc).decode())
        self._h = h
        self.device_id = int(device_id)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hadi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- introspection ------------------------------------------------------------------------
    def _raise(self, rc):
        raise HadiError(rc, self._lib.hadi_last_error(self._h).decode() or self._lib.hadi_status_string(rc).decode())

    def set_profiling(self, enabled):
        self._lib.hadi_set_profiling(self._h, 1 if enabled else 0)

    def set_tuning(self, key, value):
        """Execution-path switch (hadi.h: 'small_grid', 'small_seq', 'graph', 'american_p', 'strip', 'row_tile', 'col_groups',
        'small_waves', 'device_vgrid'); results agree to round-off."""
        rc = self._lib.hadi_set_tuning(self._h, key.encode(), int(value))
        if rc != nat.HADI_OK:
            self._raise(rc)

    def get_tuning(self, key):
        v = C.c_int()
        rc = self._lib.hadi_get_tuning(self._h, key.encode(), C.byref(v))
        if rc != nat.HADI_OK:
            raise HadiError(rc, "unknown tuning key %r" % key)
        return v.value

    def wait_stream(self, stream=None):
        """Orders the handle's stream after everything enqueued so far on `stream` (a torch.cuda.Stream; None = torch's
        current stream on this handle's device).  Called by every launcher that receives CUDA tensors, so a tensor
        written by an earlier torch op (copy_, fill_ ...) is complete before the library reads it."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(torch.device("cuda", self.device_id))
        rc = self._lib.hadi_wait_stream(self._h, C.c_void_p(stream.cuda_stream))
        if rc != nat.HADI_OK:
            self._raise(rc)

    def timing(self):
        t = nat.Timing()
        self._lib.hadi_get_timing(self._h, C.byref(t))
        return {k: getattr(t, k) for k, _ in nat.Timing._fields_}

    def device_info(self):
        name, arch, cu = C.create_string_buffer(256), C.create_string_buffer(64), C.c_int()
        self._lib.hadi_device_info(self._h, name, 256, C.byref(cu), arch, 64)
        return {"name": name.value.decode(), "arch": arch.value.decode(), "compute_units": cu.value}

    def describe_last_sweep(self):
        """Kernel names and tile geometry of the last sweep (for reports)."""
        buf = C.create_string_buffer(512)
        self._lib.hadi_describe_last_sweep(self._h, buf, 512)
        return buf.value.decode()

    # ---- problem assembly ---------------------------------------------------------------------
    def _problem(self, variant, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, grids,
                 U=None, U_0=None, lambda_bar=None, dividends=None, per_instance=None, need_vgrid=True, scheme=0,
                 state_precision=0, option_type=CALL, strikes=None):
        n = grids.Vec_s.shape[0]
        m = (m1 + 1) * (m2 + 1)
        p = nat.Problem()
        keep = []
        spaces = []

        def arr(x, name, count):
            ptr, dev = _check_array(x, name, count)
            spaces.append(dev)
            keep.append(x)
            return ptr

        p.n_instances, p.m1, p.m2, p.variant = n, m1, m2, variant
        p.scheme = int(scheme)
        p.state_precision = int(state_precision)
        p.N, p.delta_t, p.theta = int(N), float(delta_t), float(theta)
        p.r_d, p.r_f = float(r_d), float(r_f)
        p.rho, p.sigma, p.kappa, p.eta = float(rho), float(sigma), float(kappa), float(eta)
        p.vec_s = arr(grids.Vec_s, "Vec_s", n * (m1 + 1))
        p.delta_s = arr(grids.Delta_s, "Delta_s", n * m1)
        if need_vgrid:
            p.vec_v = arr(grids.Vec_v, "Vec_v", n * (m2 + 1))
            p.delta_v = arr(grids.Delta_v, "Delta_v", n * m2)
        if U is not None:
            p.U = arr(U, "U", n * m)
        if U_0 is not None:
            p.U_0 = arr(U_0, "U_0", n * m)
        if lambda_bar is not None:
            p.lambda_bar = arr(lambda_bar, "lambda_bar", n * m)
        if len(set(spaces)) > 1:
            raise ValueError("array arguments mix host and device memory")
        p.memspace = nat.MEM_DEVICE if spaces and spaces[0] else nat.MEM_HOST
        if p.memspace == nat.MEM_DEVICE:
            self.wait_stream()  # input ordering: the library's stream does not wait for torch's by itself (hadi.h)
        p.option_type = int(option_type)
        if strikes is not None:
            a = _host_f64(strikes).reshape(-1)
            if a.size != n:
                raise ValueError("strikes must have n_instances entries")
            keep.append(a)
            p.strike_i = a.ctypes.data_as(_dp)
        elif option_type == PUT:
            raise ValueError("option_type=PUT needs the strikes (boundary value K e^{-r_d t})")
        if dividends is not None and len(dividends):
            p.num_dividends = len(dividends)
            p.dividend_dates = dividends.dates.ctypes.data_as(_dp)
            p.dividend_amounts = dividends.amounts.ctypes.data_as(_dp)
            p.dividend_percentages = dividends.percentages.ctypes.data_as(_dp)
            keep.append(dividends)
        if per_instance:
            for key in ("rho_i", "sigma_i", "kappa_i", "eta_i", "delta_t_i", "V_0_i"):
                if per_instance.get(key) is not None:
                    a = _host_f64(per_instance[key])
                    if a.size != n:
                        raise ValueError("%s must have n_instances entries" % key)
                    keep.append(a)
                    setattr(p, key, a.ctypes.data_as(_dp))
            if per_instance.get("N_i") is not None:
                a = np.ascontiguousarray(np.asarray(per_instance["N_i"], dtype=np.int32))
                if a.size != n:
                    raise ValueError("N_i must have n_instances entries")
                keep.append(a)
                p.N_i = a.ctypes.data_as(_ip)
        p._keep = keep
        return p

    @staticmethod
    def _option(per_instance):
        """The launchers keep the reference's argument lists; the put extension travels in `per_instance`
        ({'option_type': PUT, 'strikes': [...]}) next to the other per-instance overrides."""
        if not per_instance:
            return CALL, None
        return per_instance.get("option_type", CALL), per_instance.get("strikes")

    def _out(self, n, cols, like):
        if _is_device(like):
            import torch
            shape = (n,) if cols == 1 else (n, cols)
            t = torch.empty(shape, dtype=torch.float64, device=like.device)
            return t, C.c_void_p(t.data_ptr())
        a = np.empty((n,) if cols == 1 else (n, cols))
        return a, C.c_void_p(a.ctypes.data)

    # ---- device_DO_timestepping* (src/device_solver.hpp:194-942) -------------------------------
    def DO_timestepping(self, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, grids, U,
                        variant=EU, U_0=None, lambda_bar=None, dividends=None, per_instance=None, scheme=0,
                        state_precision=0, option_type=CALL, strikes=None):
        """Boundary init + operator build + N Douglas steps on the caller's grids; U is updated in
        place (initial condition in, solution at T out).  scheme=1 runs Craig-Sneyd (European only);
        state_precision=1 keeps the state between the two passes in fp32 (European Douglas only, arithmetic stays fp64)."""
        p = self._problem(variant, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, grids,
                          U=U, U_0=U_0, lambda_bar=lambda_bar, dividends=dividends, per_instance=per_instance,
                          scheme=scheme, state_precision=state_precision, option_type=option_type, strikes=strikes)
        rc = self._lib.hadi_DO_timestepping(self._h, C.byref(p))
        if rc != nat.HADI_OK:
            self._raise(rc)
        return U

    # ---- diagnostics: the two directional passes of one Douglas step as operators (hadi.h) -------------------------
    def debug_row_pass(self, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, grids, U, step=1, variant=EU,
                       U_0=None, option_type=CALL, strikes=None):
        """Right-hand side of the A2 solve after the row pass of time step `step` started from U (not modified)."""
        p = self._problem(variant, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, grids, U=U, U_0=U_0,
                          option_type=option_type, strikes=strikes)
        out, optr = self._out(grids.Vec_s.shape[0], (m1 + 1) * (m2 + 1), U)
        rc = self._lib.hadi_debug_row_pass(self._h, C.byref(p), int(step), optr)
        if rc != nat.HADI_OK:
            self._raise(rc)
        return out

    def debug_col_solve(self, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, grids, rhs):
        """(I - theta dt A2)^{-1} rhs through the product's column pass (rhs not modified)."""
        p = self._problem(EU, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, grids, U=rhs)
        out, optr = self._out(grids.Vec_s.shape[0], (m1 + 1) * (m2 + 1), rhs)
        rc = self._lib.hadi_debug_col_solve(self._h, C.byref(p), optr)
        if rc != nat.HADI_OK:
            self._raise(rc)
        return out

    def debug_rcp(self, x):
        """1/x exactly as the line solves of the sweep form it (v_rcp_f64 + one Newton step), elementwise (tests)."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        out = np.empty_like(x)
        rc = self._lib.hadi_debug_rcp(self._h, int(x.size), x.ctypes.data_as(nat._dp), out.ctypes.data_as(nat._dp))
        if rc != nat.HADI_OK:
            self._raise(rc)
        return out

    # ---- CS_scheme_shuffled (src/solver.hpp:781-907), batched on the device -----------------------
    def CS_scheme(self, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, grids, U, per_instance=None):
        """Craig-Sneyd time stepping of European options: Douglas predictor + corrector that re-adds half of
        the explicit mixed-derivative increment (the reference has it in its host operator family only)."""
        return self.DO_timestepping(m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, grids, U,
                                    per_instance=per_instance, scheme=nat.SCHEME_CRAIG_SNEYD)

    # ---- parallel_DO_solve (src/device_solver.hpp:52-185) --------------------------------------
    def parallel_DO_solve(self, nInstances, S_0, V_0, m1, m2, N, T, delta_t, theta, r_d, r_f, rho, sigma,
                          kappa, eta, deviceGrids, workspace, base_prices=None):
        """European sweep + price pick.  (The reference declares S_0, V_0 as `int` here,
        device_solver.hpp:56-57, which truncates V_0 = 0.04 to 0; they are doubles in this mirror.)"""
        if deviceGrids.Vec_s.shape[0] != nInstances:
            raise ValueError("nInstances does not match the grid batch")
        p = self._problem(EU, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, deviceGrids,
                          U=workspace.U)
        out, optr = self._out(nInstances, 1, workspace.U)
        rc = self._lib.hadi_parallel_DO_solve(self._h, C.byref(p), float(S_0), float(V_0), optr)
        if rc != nat.HADI_OK:
            self._raise(rc)
        if base_prices is not None:
            base_prices[...] = out
            return base_prices
        return out

    # ---- compute_base_prices* (src/jacobian_computation.cpp:368, 629, 922, 1232) ---------------
    def _base_prices(self, variant, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N,
                     theta, delta_t, num_strikes, deviceGrids, workspace, U_0=None, dividends=None,
                     per_instance=None):
        option_type, strikes = self._option(per_instance)
        if total_size != (m1 + 1) * (m2 + 1):
            raise ValueError("total_size != (m1+1)*(m2+1)")
        if deviceGrids.Vec_s.shape[0] != num_strikes:
            raise ValueError("num_strikes does not match the grid batch")
        p = self._problem(variant, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, deviceGrids,
                          U=workspace.U, U_0=U_0, lambda_bar=None, dividends=dividends,
                          per_instance=per_instance, need_vgrid=False, option_type=option_type, strikes=strikes)
        out, optr = self._out(num_strikes, 1, workspace.U)
        rc = self._lib.hadi_compute_base_prices(self._h, C.byref(p), float(S_0), float(V_0), optr)
        if rc != nat.HADI_OK:
            self._raise(rc)
        return out

    def compute_base_prices(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                            delta_t, num_strikes, deviceGrids, workspace, per_instance=None):
        return self._base_prices(EU, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                                 delta_t, num_strikes, deviceGrids, workspace, per_instance=per_instance)

    def compute_base_prices_american(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N,
                                     theta, delta_t, num_strikes, deviceGrids, U_0, workspace, per_instance=None):
        return self._base_prices(AM, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                                 delta_t, num_strikes, deviceGrids, workspace, U_0=U_0, per_instance=per_instance)

    def compute_base_prices_dividends(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N,
                                      theta, delta_t, num_strikes, deviceGrids, workspace, dividends,
                                      per_instance=None):
        return self._base_prices(DIV, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                                 delta_t, num_strikes, deviceGrids, workspace, dividends=dividends,
                                 per_instance=per_instance)

    def compute_base_prices_american_dividends(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2,
                                               total_size, N, theta, delta_t, num_strikes, deviceGrids, U_0,
                                               workspace, dividends, per_instance=None):
        return self._base_prices(AM_DIV, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N,
                                 theta, delta_t, num_strikes, deviceGrids, workspace, U_0=U_0, dividends=dividends,
                                 per_instance=per_instance)

    # ---- compute_jacobian* (src/jacobian_computation.cpp:204, 457, 726, 1031) ------------------
    def _jacobian(self, variant, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                  delta_t, num_strikes, deviceGrids, U_0, eps, dividends=None, per_instance=None):
        if total_size != (m1 + 1) * (m2 + 1):
            raise ValueError("total_size != (m1+1)*(m2+1)")
        option_type, strikes = self._option(per_instance)
        p = self._problem(variant, m1, m2, N, delta_t, theta, r_d, r_f, rho, sigma, kappa, eta, deviceGrids,
                          U=None, U_0=U_0, dividends=dividends, per_instance=per_instance, need_vgrid=False,
                          option_type=option_type, strikes=strikes)
        J, jptr = self._out(num_strikes, 5, U_0)
        base, bptr = self._out(num_strikes, 1, U_0)
        rc = self._lib.hadi_compute_jacobian(self._h, C.byref(p), float(S_0), float(V_0), float(eps), jptr, bptr)
        if rc != nat.HADI_OK:
            self._raise(rc)
        return J, base

    def compute_jacobian(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                         delta_t, num_strikes, deviceGrids, U_0, eps=1e-6, per_instance=None):
        """Returns (J [n][5] with columns kappa, eta, sigma, rho, v0; base_prices [n])."""
        return self._jacobian(EU, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                              delta_t, num_strikes, deviceGrids, U_0, eps, per_instance=per_instance)

    def compute_jacobian_american(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N,
                                  theta, delta_t, num_strikes, deviceGrids, U_0, eps=1e-6, per_instance=None):
        return self._jacobian(AM, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                              delta_t, num_strikes, deviceGrids, U_0, eps, per_instance=per_instance)

    def compute_jacobian_dividends(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N,
                                   theta, delta_t, num_strikes, deviceGrids, U_0, dividends, eps=1e-6,
                                   per_instance=None):
        return self._jacobian(DIV, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                              delta_t, num_strikes, deviceGrids, U_0, eps, dividends=dividends,
                              per_instance=per_instance)

    def compute_jacobian_american_dividends(self, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2,
                                            total_size, N, theta, delta_t, num_strikes, deviceGrids, U_0,
                                            dividends, eps=1e-6, per_instance=None):
        return self._jacobian(AM_DIV, S_0, V_0, T, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                              delta_t, num_strikes, deviceGrids, U_0, eps, dividends=dividends,
                              per_instance=per_instance)

    # ---- multi-maturity launchers (src/heston_calibration.cpp:2174-2424, 2936-3243) ----------------------
    # One instance per CalibrationPoint{strike, maturity, time_steps, delta_t, global_index}: the batch shares the
    # grid shape and the model parameters, every instance steps its own (N, delta_t).
    @staticmethod
    def _steps(calibration_points, total_calibration_size):
        if len(calibration_points) != total_calibration_size:
            raise ValueError("total_calibration_size does not match the calibration points")
        N_i = np.array([pt.time_steps for pt in calibration_points], dtype=np.int32)
        dt_i = np.array([pt.delta_t for pt in calibration_points], dtype=np.float64)
        return {"N_i": N_i, "delta_t_i": dt_i}, int(N_i.max()) if len(N_i) else 1, float(dt_i[0]) if len(dt_i) else 1.0

    def compute_jacobian_multi_maturity(self, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, theta,
                                        calibration_points, total_calibration_size, deviceGrids, U_0, eps=1e-6):
        per, N, dt = self._steps(calibration_points, total_calibration_size)
        return self._jacobian(EU, S_0, V_0, None, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta, dt,
                              total_calibration_size, deviceGrids, U_0, eps, per_instance=per)

    def compute_base_prices_multi_maturity(self, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, theta,
                                           calibration_points, total_calibration_size, deviceGrids, workspace):
        per, N, dt = self._steps(calibration_points, total_calibration_size)
        return self._base_prices(EU, S_0, V_0, None, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                                 dt, total_calibration_size, deviceGrids, workspace, per_instance=per)

    def compute_jacobian_multi_maturity_american_dividends(self, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2,
                                                           total_size, theta, calibration_points,
                                                           total_calibration_size, deviceGrids, U_0, dividends,
                                                           eps=1e-6):
        per, N, dt = self._steps(calibration_points, total_calibration_size)
        return self._jacobian(AM_DIV, S_0, V_0, None, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N, theta,
                              dt, total_calibration_size, deviceGrids, U_0, eps, dividends=dividends, per_instance=per)

    def compute_base_prices_multi_maturity_american_dividends(self, S_0, V_0, r_d, r_f, rho, sigma, kappa, eta, m1, m2,
                                                              total_size, theta, calibration_points,
                                                              total_calibration_size, deviceGrids, U_0, workspace,
                                                              dividends):
        per, N, dt = self._steps(calibration_points, total_calibration_size)
        return self._base_prices(AM_DIV, S_0, V_0, None, r_d, r_f, rho, sigma, kappa, eta, m1, m2, total_size, N,
                                 theta, dt, total_calibration_size, deviceGrids, workspace, U_0=U_0,
                                 dividends=dividends, per_instance=per)


# ---- LM linear algebra (host; jacobian_computation.cpp:20-195) -------------------------------------
def lm_partials(J, residuals):
    """[J^T J (25), J^T r (5), sum r^2 (1)] of this rank's rows: the 31 doubles that are all-reduced."""
    J, r = _host_f64(J), _host_f64(residuals)
    out = np.empty(31)
    rc = nat.lib().hadi_lm_partials(J.shape[0], J.ctypes.data_as(_dp), r.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
    if rc != nat.HADI_OK:
        raise HadiError(rc, "hadi_lm_partials")
    return out


def lm_partials_device(solver, J, model_prices, market_prices):
    """The same 31 doubles reduced on the GPU from device-resident J [n][5], model prices [n] and market prices [n]
    (CUDA tensors): only 31 doubles leave the device per LM iteration (hadi_lm_partials_device)."""
    n = int(J.shape[0])
    for t, name, cnt in ((J, "J", 5 * n), (model_prices, "model_prices", n), (market_prices, "market_prices", n)):
        _check_array(t, name, cnt)
        if not _is_device(t):
            raise ValueError("%s must be a CUDA tensor" % name)
    solver.wait_stream()
    out = np.empty(31)
    rc = solver._lib.hadi_lm_partials_device(solver._h, n, C.c_void_p(J.data_ptr()), C.c_void_p(model_prices.data_ptr()),
                                             C.c_void_p(market_prices.data_ptr()), out.ctypes.data_as(_dp))
    if rc != nat.HADI_OK:
        solver._raise(rc)
    return out


def lm_solve(partials31, lam):
    part = _host_f64(partials31)
    delta = np.empty(5)
    rc = nat.lib().hadi_lm_solve(part.ctypes.data_as(_dp), float(lam), delta.ctypes.data_as(_dp))
    if rc != nat.HADI_OK:
        raise HadiError(rc, "hadi_lm_solve")
    return delta


def compute_parameter_update(J, residuals, lam):
    """compute_parameter_update_on_device (jacobian_computation.cpp:107-195)."""
    return lm_solve(lm_partials(J, residuals), lam)
