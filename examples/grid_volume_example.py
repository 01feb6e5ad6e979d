#!/usr/bin/env python3
"""End-to-end use of the MI355X path with the reference's call sequence
(examples/radar_grid_building_example.py + radar_grid_interpolation_example.py + radar_grid_products_example.py of
jgmarti84/radar-processor): build the geometry once, save / load it, grid fields with a QC filter, derive products.

A synthetic volume stands in for ``pyart.io.read(...)`` (PyART is not needed by this package; any object with the
duck-typed Radar attributes works).  Needs an MI355X and the built library.

    python examples/grid_volume_example.py [--config C2]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from radar_processor_amd import (GateFilter, apply_colormap_to_array, apply_filter_masks, apply_geometry,  # noqa: E402
                                 apply_geometry_multi, collapse_field_3d_to_2d, column_argmax, column_max,
                                 compute_grid_geometry, constant_altitude_ppi, constant_elevation_ppi,
                                 get_field_data, get_gate_coordinates, get_radar_info, load_geometry, save_geometry)
from radar_processor_amd import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2", choices=sorted(synthetic.CONFIGS))
    args = ap.parse_args()
    cfg = synthetic.CONFIGS[args.config]

    radar = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0,
                                  fields=("DBZH", "ZDR", "RHOHV")).as_radar()
    print(get_radar_info(radar))

    # 1. geometry: once per radar + scan strategy + grid (reference: minutes to hours; here: a fraction of a second)
    gate_x, gate_y, gate_z = get_gate_coordinates(radar)
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.time()
        geometry = compute_grid_geometry(gate_x, gate_y, gate_z, cfg["grid_shape"], cfg["grid_limits"], temp_dir=tmp,
                                         toa=17000.0)
        print(f"geometry built in {time.time() - t0:.2f} s\n{geometry}")
        if geometry.n_pairs() < 5e7:          # same .npz format as radar_grid.save_geometry
            path = os.path.join(tmp, "geometry.npz")
            save_geometry(geometry, path)
            geometry = load_geometry(path)

    # 2. per volume: QC filter + gridding (one fused CSR pass for the three fields)
    gf = GateFilter(radar).exclude_below("RHOHV", 0.8)
    print(gf.summary())
    t0 = time.time()
    dbzh = apply_geometry(geometry, get_field_data(radar, "DBZH"), additional_filters=[gf])
    print(f"apply_geometry: {time.time() - t0 :.3f} s -> {dbzh.shape} {dbzh.dtype}, {np.isfinite(dbzh).mean():.1%} filled")
    t0 = time.time()
    grids = apply_geometry_multi(geometry, {n: get_field_data(radar, n) for n in ("DBZH", "ZDR", "RHOHV")},
                                 additional_filters={n: [gf] for n in ("DBZH", "ZDR")})
    print(f"apply_geometry_multi (3 fields): {time.time() - t0:.3f} s")

    # 3. products
    cappi = constant_altitude_ppi(grids["DBZH"], geometry, 4000.0)
    colmax, level = column_argmax(grids["DBZH"])
    ppi = constant_elevation_ppi(grids["DBZH"], geometry, 0.9)
    print(f"CAPPI@4000 m max {np.nanmax(cappi):.1f} dBZ | COLMAX max {np.nanmax(colmax):.1f} dBZ at level "
          f"{int(level.ravel()[np.nanargmax(colmax)])} | PPI 0.9 deg: {np.isfinite(ppi).mean():.1%} of pixels inside the grid")
    assert np.array_equal(colmax, column_max(grids["DBZH"]), equal_nan=True)

    # 4. the 2-D tier of process_radar_to_cog: processor-style collapse, visual + QC filter masks, colormap -> RGBA
    class Range:                                     # what the processor's filter objects look like
        def __init__(self, field, lo, hi):
            self.field, self.min, self.max = field, lo, hi

    nz, ny, nx = geometry.grid_shape
    (z_lo, z_hi), (y_lo, y_hi), (x_lo, x_hi) = geometry.grid_limits
    axes = dict(x_coords=np.linspace(x_lo, x_hi, nx), y_coords=np.linspace(y_lo, y_hi, ny),
                z_levels=np.linspace(z_lo, z_hi, nz))
    planes = {n: collapse_field_3d_to_2d(np.ma.masked_invalid(grids[n]), "ppi", elevation_deg=0.9, **axes)
              for n in ("DBZH", "RHOHV")}
    shown = apply_filter_masks(planes["DBZH"], [Range("DBZH", -10.0, None)], [Range("RHOHV", 0.85, None)], "DBZH",
                               {"qc": {"RHOHV": planes["RHOHV"]}})
    rgba = apply_colormap_to_array(shown.filled(np.nan), "turbo", vmin=-10.0, vmax=70.0)
    print(f"processor-style PPI: {shown.count()} of {shown.size} pixels shown -> RGBA {rgba.shape} {rgba.dtype}, "
          f"{(rgba[..., 3] == 0).mean():.1%} transparent")


if __name__ == "__main__":
    main()
