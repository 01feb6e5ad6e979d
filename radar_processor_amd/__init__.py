"""radar_processor_amd -- MI355X-native implementation of the ``radar_grid`` hot path of
jgmarti84/radar-processor: geometry build -> CSR apply -> CAPPI / COLMAX, behind the reference's own Python
function surface (``radar_grid/__init__.py:39-82``, hot-path names only).

    from radar_processor_amd import (GridGeometry, compute_grid_geometry, apply_geometry, apply_geometry_multi,
                                     GateFilter, constant_altitude_ppi, column_max, save_geometry, load_geometry)

Host code is Python on PyTorch-ROCm tensors (allocations + streams only); the work is done by hand-written HIP
kernels for gfx950 in ``csrc/`` behind the C ABI of ``include/radargrid_hip.h``.  There is no CPU fallback:
without the built library and a HIP device every compute entry point raises ``NativeUnavailable``.

(The directory is spelled ``radar_processor_amd`` because a hyphen cannot appear in a Python package name;
``radar-processor_amd`` at the repository root is a symlink to it.)
"""
from ._native import NativeError, NativeUnavailable, load_library
from .gate_filters import GateFilter, GridFilter, create_mask_from_filter, device_gate_mask
from .geometry_builder import RoiSearch, compute_grid_geometry
from .grid_geometry import (DeviceCSR, GridGeometry, load_device_layout, load_geometry, save_device_layout, save_geometry)
from .grid_products import (EARTH_RADIUS, EFFECTIVE_RADIUS_FACTOR, column_argmax, column_max, column_mean,
                            column_min, compute_beam_height, compute_beam_height_flat, compute_beam_height_simple,
                            constant_altitude_ppi, constant_elevation_ppi, get_beam_height_difference,
                            get_elevation_from_z_level)
from .gridding import PlaneProducts, apply_geometry, apply_geometry_multi, grid_fields_device, grid_products_device
from .roi_grid import roi_grid_fields_device
from .processor_seam import build_grid3d_package
from .raster import (PlaneTest, apply_colormap_to_array, apply_filter_masks, collapse_field_3d_to_2d,
                     collapse_grid_to_2d, collapse_plane_device, colormap_lut, colormap_rgba_device,
                     plane_filter_device)
from .radar_adaptors import (get_available_fields, get_field_data, get_gate_coordinates, get_radar_altitude,
                             get_radar_info)

__version__ = "0.1.0"

__all__ = [
    # reference surface (radar_grid/__init__.py:39-82), hot-path subset
    "GridGeometry", "save_geometry", "load_geometry",
    "compute_grid_geometry",
    "apply_geometry", "apply_geometry_multi",
    "get_gate_coordinates", "get_field_data", "get_available_fields", "get_radar_info", "get_radar_altitude",
    "GateFilter", "GridFilter", "create_mask_from_filter",
    "constant_altitude_ppi", "constant_elevation_ppi", "column_max", "column_min", "column_mean",
    "get_elevation_from_z_level", "get_beam_height_difference", "compute_beam_height", "compute_beam_height_flat",
    "EARTH_RADIUS", "EFFECTIVE_RADIUS_FACTOR", "apply_colormap_to_array",
    # 2-D raster stage of radar_processor (utils.py:336-387, processor.py:480-551, :802-886)
    "collapse_field_3d_to_2d", "collapse_grid_to_2d", "apply_filter_masks",
    # build-specific additions
    "save_device_layout", "load_device_layout", "column_argmax", "grid_fields_device", "grid_products_device", "PlaneProducts", "roi_grid_fields_device", "build_grid3d_package", "device_gate_mask", "RoiSearch", "DeviceCSR",
    "collapse_plane_device", "plane_filter_device", "PlaneTest", "colormap_lut", "colormap_rgba_device",
    "NativeUnavailable", "NativeError", "load_library",
]
