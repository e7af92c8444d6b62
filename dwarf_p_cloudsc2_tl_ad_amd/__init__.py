"""MI355X-native CLOUDSC2 column-physics engine (nonlinear, tangent-linear, adjoint).

Only the hot path of ecmwf-ifs/dwarf-p-cloudsc2-tl-ad lives here: hand-written HIP kernels for gfx950 behind a C ABI
(include/cloudsc2_hip.h), the Fortran drivers with the reference's signatures (fortran/) and this Python mirror of
the same driver interface.  Importing the package loads the HIP library and raises if it has not been built -- there
is no CPU fallback.
"""
from . import binding  # noqa: F401  (raises ImportError when libcloudsc2_hip.so is missing)
from .binding import (Cloudsc2Error, Params, adjoint_verdict, default_params, device_available, get_math_mode,  # noqa: F401
                      set_math_mode, taylor_verdict)
from .state import (Cloudsc2State, bytes_per_column, ceta_from_table, column_range, nblocks_of, random_table,  # noqa: F401
                    state_from_table, synthetic_table, validate_l1)
from .driver import (DeviceState, FlatFields, ResidentState, cloudsc_driver, cloudsc_driver_ad, cloudsc_driver_tl,  # noqa: F401
                     run_state)
