"""MI355X-native batched Heston Douglas-ADI time stepper (libhadi) -- Python host mirror.

Drop-in for the hot path of BCW-dot/PDE-based-Heston-Solver-GPU-accelerated:
`parallel_DO_solve`, `compute_base_prices*`, `compute_jacobian*`, `device_DO_timestepping*`.
The compute runs in hand-written HIP kernels for gfx950 behind the C ABI in include/hadi.h;
this package only marshals arguments.  There is no CPU fallback.
"""
from ._native import EU, AM, DIV, AM_DIV, CALL, PUT, HadiError, LIB_PATH, STATE_FP64, STATE_FP32  # noqa: F401
from .grid import Grid, GridViewsBatch  # noqa: F401
from .solver import (HestonADI, DOWorkspace, Dividends, compute_parameter_update,  # noqa: F401
                     lm_partials, lm_partials_device, lm_solve)
from .distributed import Communicator, shard_range  # noqa: F401
from .calibration import (CalibrationPoint, calibrate, calibrate_american, calibrate_american_dividends,  # noqa: F401
                          calibrate_american_dividends_multi_maturity, calibrate_dividends, calibrate_european,
                          calibrate_european_multi_maturity, clamp_parameters, export_calibration_csv,
                          make_calibration_points)
from . import market  # noqa: F401

__all__ = ["EU", "AM", "DIV", "AM_DIV", "CALL", "PUT", "lm_partials_device", "STATE_FP64", "STATE_FP32", "HadiError", "Grid", "GridViewsBatch", "HestonADI", "DOWorkspace",
           "Dividends", "compute_parameter_update", "lm_partials", "lm_solve", "LIB_PATH", "Communicator",
           "shard_range", "calibrate_european", "clamp_parameters", "market", "CalibrationPoint", "calibrate",
           "calibrate_american", "calibrate_dividends", "calibrate_american_dividends",
           "calibrate_european_multi_maturity", "calibrate_american_dividends_multi_maturity",
           "make_calibration_points", "export_calibration_csv"]
