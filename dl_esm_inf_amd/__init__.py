"""MI355X-native hot path of dl_esm_inf: r2d_field stencil updates + depth-1 halo exchange.

Layout
  csrc/      HIP kernels + the C-ABI library (include/dlesm_hip.h -> lib/libdlesm_hip.so)
  fortran/   the drop-in Fortran API layer (same module / type / procedure names as the
             reference) that binds the C ABI through ISO_C_BINDING
  *.py       a thin Python mirror of the same interface (ctypes over the same C ABI) used by
             tests/ and bench.py; torch is only the device-memory / stream provider

Nothing here falls back to the CPU: without the built library the import fails, without a GPU
every device entry point raises.
"""
from . import _cabi
from ._cabi import DlesmError, GoceanStop  # noqa: F401

_cabi.lib()  # fail loudly, at import time, if the HIP extension is missing

from .parallel_mod import (parallel_init, parallel_finalise, get_rank, get_num_ranks,  # noqa: E402,F401
                           on_master, go_decompose, map_comms)
from .grid_mod import (grid_type, grid_init, GO_ARAKAWA_C, GO_ARAKAWA_B, GO_OFFSET_SW,  # noqa: E402,F401
                       GO_OFFSET_SE, GO_OFFSET_NW, GO_OFFSET_NE, GO_OFFSET_ANY,
                       GO_BC_PERIODIC, GO_BC_EXTERNAL, GO_BC_NONE)
from .field_mod import (r2d_field, field_checksum, copy_field, set_field, free_field,  # noqa: E402,F401
                        GO_U_POINTS, GO_V_POINTS, GO_T_POINTS, GO_F_POINTS, GO_ALL_POINTS)
from . import psy  # noqa: E402,F401
