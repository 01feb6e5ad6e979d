"""PyART adaptors -- host-side mirror of ``radar_grid/utils.py`` (:12-130).

These only re-shape what a (duck-typed) ``pyart.core.Radar`` already holds; they are inputs to the hot path,
not part of it.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def get_gate_coordinates(radar) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Flat float32 ``(gate_x, gate_y, gate_z)`` in metres relative to the radar (``utils.py:35-38``)."""
    return tuple(getattr(radar, name)["data"].ravel().astype("float32") for name in ("gate_x", "gate_y", "gate_z"))


def get_field_data(radar, field_name: str) -> np.ndarray:
    """Flat float32 masked array with NaN/Inf masked (``utils.py:64-66``)."""
    return np.ma.masked_invalid(radar.fields[field_name]["data"]).ravel().astype("float32")


def get_available_fields(radar) -> list:
    return list(radar.fields.keys())


def get_radar_altitude(radar) -> float:
    return float(radar.altitude["data"][0])


def get_radar_info(radar) -> dict:
    """Metadata summary with the reference's keys (``utils.py:116-130``)."""
    md = radar.metadata
    return {
        "radar_name": md.get("instrument_name", "UNKNOWN"),
        "strategy": md.get("scan_id", "UNKNOWN"),
        "volume_nr": f"{int(md.get('volume_number', 0)):02d}",
        "nrays": radar.nrays,
        "ngates": radar.ngates,
        "nsweeps": radar.nsweeps,
        "total_gates": radar.nrays * radar.ngates,
        "fields": list(radar.fields.keys()),
        "range_min": float(radar.range["data"][0]),
        "range_max": float(radar.range["data"][-1]),
        "latitude": float(radar.latitude["data"][0]),
        "longitude": float(radar.longitude["data"][0]),
        "altitude": float(radar.altitude["data"][0]),
    }
