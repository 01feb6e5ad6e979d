#!/usr/bin/env python3
"""Event timings of the 2-D raster stage (collapse / filter masks / colormap) on a 40 x 2000 x 2000 grid; quoted in
DESIGN.md.  Everything stays in HBM."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import radar_processor_amd as rg
    dev = torch.device("cuda")
    nz, ny, nx = 40, 2000, 2000
    gen = torch.Generator(device=dev).manual_seed(0)
    grid = torch.randn((nz, ny, nx), device=dev, generator=gen) * 15 + 20
    grid[torch.rand((nz, ny, nx), device=dev, generator=gen) < 0.3] = float("nan")
    rho = torch.rand((ny, nx), device=dev, generator=gen)
    x = np.linspace(-240e3, 240e3, nx); y = np.linspace(-240e3, 240e3, ny); z = np.linspace(0, 15e3, nz)
    lut = torch.from_numpy(rg.colormap_lut("turbo")).to(dev)

    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            out = fn()
        b.record(); b.synchronize()
        return a.elapsed_time(b) / reps * 1e3, out   # microseconds

    rec = {}
    rec["collapse_ppi_us"], plane = timed(lambda: rg.collapse_plane_device(grid, "ppi", x_coords=x, y_coords=y, z_levels=z,
                                                                        elevation_deg=1.3))
    rec["collapse_colmax_us"], cm = timed(lambda: rg.collapse_plane_device(grid, "colmax"))
    rec["collapse_cappi_us"], _ = timed(lambda: rg.collapse_plane_device(grid, "cappi", z_levels=z, target_height_m=3000.0))
    tests = [rg.PlaneTest(lo=-30.0, lo_inclusive=True, nonfinite=True), rg.PlaneTest(hi=60.0), rg.PlaneTest(plane=rho, lo=0.8)]
    rec["plane_filter_3tests_us"], (vals, _) = timed(lambda: rg.plane_filter_device(cm, tests))
    rec["colormap_auto_limits_us"], _ = timed(lambda: rg.colormap_rgba_device(vals, lut))
    rec["colormap_fixed_limits_us"], rgba = timed(lambda: rg.colormap_rgba_device(vals, lut, -10.0, 70.0))
    n = ny * nx
    rec["colormap_fixed_GBps"] = round(8 * n / rec["colormap_fixed_limits_us"] / 1e3, 1)
    rec["collapse_colmax_GBps"] = round(4 * (nz + 1) * n / rec["collapse_colmax_us"] / 1e3, 1)
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in rec.items()}), flush=True)


if __name__ == "__main__":
    main()
