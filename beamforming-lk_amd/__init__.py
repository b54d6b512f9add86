"""beamforming-lk_amd -- MI355X (gfx950) delay-and-sum heatmap engine behind the
aw_processing_unit API of acoustic-warfare/beamforming-lk.

The product is the C-ABI library libawpu_hip.so (include/awpu_hip.h, sources in csrc/)
and the C++ host mirror of the reference's MIMO worker (host/).  This Python package is
plumbing: it builds and loads the library (ctypes) for the tests and bench.py.

The directory name has a hyphen; import it with
    importlib.import_module("beamforming-lk_amd")
"""
from . import _build, binding, synthetic  # noqa: F401
from .binding import (  # noqa: F401
    MATH_BF16_ACC,
    MATH_F32_EXACT,
    MATH_F32_FAST,
    AwpuError,
    Engine,
    build_delay_table,
    build_delay_table_device,
    create_antenna,
    create_tiled_antenna,
    heatmap_u8,
    resize_linear_u8,
    steer_table,
    steering_delays,
)

__all__ = [
    "Engine", "AwpuError", "MATH_F32_EXACT", "MATH_F32_FAST", "MATH_BF16_ACC", "build_delay_table", "build_delay_table_device",
    "create_antenna", "create_tiled_antenna", "steering_delays", "heatmap_u8", "resize_linear_u8", "steer_table", "binding",
    "synthetic", "_build",
]
