#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md (not the bench contract): PCIe-inclusive apply_geometry, geometry
build time, fused no-CSR gridder (K2) rate, K1-vs-K2 agreement at full size."""
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    out = {}
    which = sys.argv[1:] or ["C2", "METRIC"]
    for name in which:
        cfg = synthetic.CONFIGS[name]
        vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
        shape, limits = cfg["grid_shape"], cfg["grid_limits"]
        n_vox = int(np.prod(shape))
        rec = {}
        torch.cuda.synchronize()
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.perf_counter()
            geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, tmp)
            torch.cuda.synchronize()
            rec["geometry_build_s"] = round(time.perf_counter() - t0, 3)
        rec["pairs"] = geom.n_pairs()
        field = vol.fields["DBZH"]
        rg.apply_geometry(geom, field)                       # warm
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            g_host = rg.apply_geometry(geom, field)
            ts.append(time.perf_counter() - t0)
        rec["apply_geometry_host_inclusive_ms"] = round(min(ts) * 1e3, 2)
        rec["apply_geometry_host_inclusive_mvoxel_s"] = round(n_vox / min(ts) / 1e6, 1)
        # fused K2
        search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits)
        dev = search.dev
        f_t = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(field))).to(dev)
        m_t = torch.from_numpy(np.ma.getmaskarray(field).astype(np.uint8)).to(dev)
        grid = rg.roi_grid_fields_device(search, [f_t], [m_t])
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        grid = rg.roi_grid_fields_device(search, [f_t], [m_t], out=grid)
        b.record()
        b.synchronize()
        rec["roi_grid_fused_ms"] = round(a.elapsed_time(b), 2)
        rec["roi_grid_fused_mvoxel_s"] = round(n_vox / a.elapsed_time(b) / 1e3, 1)
        k2 = grid[0].cpu().numpy()
        both = np.isfinite(k2) & np.isfinite(g_host)
        rec["k1_vs_k2_nan_pattern_equal"] = bool(np.array_equal(np.isnan(k2), np.isnan(g_host)))
        rec["k1_vs_k2_max_abs_diff"] = float(np.abs(k2[both] - g_host[both]).max())
        rec["filled_voxel_fraction"] = round(float(both.mean()), 4)
        out[name] = rec
        print(name, json.dumps(rec), flush=True)
        del geom, search, grid
        torch.cuda.empty_cache()
    if "C4" in sys.argv[1:] or not sys.argv[1:]:
        cfg = synthetic.CONFIGS["C4"]
        vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=4, fields=("DBZH",))
        shape, limits = cfg["grid_shape"], cfg["grid_limits"]
        t0 = time.perf_counter()
        search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits)
        torch.cuda.synchronize()
        rec = {"roi_search_setup_s": round(time.perf_counter() - t0, 3), "cell_size_m": round(search.cell_size, 1)}
        f_t = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vol.fields["DBZH"]))).to(search.dev)
        m_t = torch.from_numpy(np.ma.getmaskarray(vol.fields["DBZH"]).astype(np.uint8)).to(search.dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        grid = rg.roi_grid_fields_device(search, [f_t], [m_t])
        b.record()
        b.synchronize()
        rec["roi_grid_fused_ms"] = round(a.elapsed_time(b), 2)
        rec["roi_grid_fused_mvoxel_s"] = round(int(np.prod(shape)) / a.elapsed_time(b) / 1e3, 1)
        rec["filled_voxel_fraction"] = round(float(torch.isfinite(grid).float().mean().item()), 4)
        out["C4_fused"] = rec
        print("C4_fused", json.dumps(rec), flush=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "measure_misc.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
