#!/usr/bin/env python3
"""Benchmark of the radar_grid hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config METRIC|C2|C4] [--fields F] [--mode csr|fused]

One *step* = one pass of the per-volume hot path over one synthetic volume per GPU, inputs resident in HBM:
mask fold + field interleave (rg_pack_fields_f32) -> gridding (rg_csr_apply_f32, the dominant kernel) ->
COLMAX+argmax (rg_column_reduce_f32) -> CAPPI@4000 m (rg_cappi_lerp_f32).  The geometry (CSR) is built once
on the GPU before the timed region -- that is how the reference uses it too (once per scan strategy,
SURVEY.md §3.1).

Default workload (N=1): the configuration BASELINE.json's metric is quoted on -- the 12-elevation
360x1000-gate volume onto the 40x2000x2000 grid, one field (DBZH).  For N>1 (launched by
torch.distributed.run, one rank per GPU) every rank grids its own volume: independent volumes shard with no
data-path collective (weak scaling); RCCL is used only for the barrier / max-over-ranks of the timing.

Rank 0 prints ONE JSON line on stdout; progress goes to stderr.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s copy)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="METRIC", help="METRIC (default), C2, C4 -- see radar_processor_amd.synthetic.CONFIGS")
    ap.add_argument("--fields", type=int, default=1, help="fields gridded per volume in one fused CSR pass (1..3)")
    ap.add_argument("--volumes-per-gpu", type=int, default=1, help="volumes fused into each step on every GPU")
    ap.add_argument("--mode", choices=("csr", "fused"), default="csr",
                    help="csr = precomputed geometry + rg_csr_apply_f32 (K1); fused = rg_roi_grid_f32 (K2, no CSR)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-compact", action="store_true",
                    help="csr mode, single field-volume: run the standard 8-byte-per-pair kernel instead of the "
                         "compact device copy of the CSR (rg_csr_compact_apply_f32, identical results)")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend for N>1 (nccl = RCCL; gloo only for rehearsing ranks on one GPU)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --dist-backend gloo; RCCL refuses a shared GPU)")
    ap.add_argument("--cpu-sample-pairs", type=float, default=1.5e9, help="upper bound on CSR pairs in the CPU sample")
    return ap.parse_args()


def cpu_baseline(geom, vol, field_name, n_vox_total, max_pairs):
    """Reference CPU path (NumPy, single thread -- what radar_grid/interpolate.py does) on a bounded sample:
    whole y-rows around the centre of the grid, CSR rows copied from the GPU-built geometry."""
    from oracle import radar_grid_oracle as oracle
    csr = geom.device_csr()
    nz, ny, nx = geom.grid_shape
    ip_rows = csr.indptr[::nx].cpu().numpy().astype(np.int64)      # pair offset at the start of every y-row
    n_rows = nz * ny
    centre = (nz // 2) * ny + ny // 2
    # widest window of whole y-rows around the grid centre that stays under the pair budget (bisection)
    h_lo, h_hi = 1, max(centre, n_rows - centre)
    while h_lo < h_hi:
        h = (h_lo + h_hi + 1) // 2
        if ip_rows[min(n_rows, centre + h)] - ip_rows[max(0, centre - h)] <= max_pairs:
            h_lo = h
        else:
            h_hi = h - 1
    r0, r1 = max(0, centre - h_lo), min(n_rows, centre + h_lo)
    v0, v1 = r0 * nx, r1 * nx
    p0, p1 = int(ip_rows[r0]), int(ip_rows[r1])
    indptr = (csr.indptr[v0:v1 + 1].cpu().numpy().astype(np.int64) - p0)
    if csr.gate_indices is not None:
        gidx = csr.gate_indices[p0:p1].cpu().numpy()
    else:                                          # compact-only geometry: rebuild the sample's index array
        gidx = geom.device_compact().decode(csr, v0, v1).cpu().numpy()
    wts = csr.weights[p0:p1].cpu().numpy()
    shape = (1, r1 - r0, nx)
    data, mask = oracle.merge_masks(vol.fields[field_name])
    oracle.csr_apply(indptr[:2], gidx[:int(indptr[1])], wts[:int(indptr[1])], data, mask, (1, 1, 1))  # touch code paths
    t0 = time.perf_counter()
    oracle.csr_apply(indptr, gidx, wts, data, mask, shape)
    dt = time.perf_counter() - t0
    pairs = p1 - p0
    pairs_per_s = pairs / dt
    total_pairs = csr.n_pairs
    full_grid_s = total_pairs / pairs_per_s
    return {
        "value": round(n_vox_total / full_grid_s / 1e6, 4),
        "unit": "Mvoxel/s",
        "cores": 1,
        "kind": "port",
        "sample": (f"NumPy restatement of apply_geometry (oracle.csr_apply, 1 thread) on y-rows [{r0},{r1}) of the "
                   f"bench grid's {n_rows} (z,y) rows around its centre: {v1 - v0} voxels, {pairs} CSR pairs in {dt:.2f} s = {pairs_per_s / 1e6:.1f} Mpairs/s; "
                   f"value = full-grid voxels / (all {total_pairs} pairs / that rate)"),
        "sample_mvoxel_per_s": round((v1 - v0) / dt / 1e6, 4),
        "mpairs_per_s": round(pairs_per_s / 1e6, 2),
    }


def pmc_traffic(workload: str):
    """HBM bytes per csr_apply launch from rocprofv3 PMC passes (profiles/pmc_traffic.json, committed with the
    rocprof CSVs it was derived from); None when no measurement exists for this workload."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
        return table.get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    from radar_processor_amd.gridding import CsrGridder

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    n_gpus = world
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group(backend="gloo")
    if local_rank == 0:
        from radar_processor_amd.build import ensure_built
        ensure_built(verbose=rank == 0)          # bare checkout: compile the git-ignored library once per node
    if world > 1:
        dist.barrier()
    rg.load_library()

    cfg = synthetic.CONFIGS[args.config]
    field_names = ("DBZH", "ZDR", "RHOHV")[:max(1, min(3, args.fields))]
    n_f = len(field_names)
    n_vol = max(1, args.volumes_per_gpu)
    if n_f * n_vol > 8:
        raise SystemExit("fields x volumes-per-gpu must be <= 8 (one fused pass)")
    shape, limits = cfg["grid_shape"], cfg["grid_limits"]
    n_vox = int(np.prod(shape))

    # ---- synthetic inputs, resident in HBM before the timed region ------------------------------------
    t0 = time.perf_counter()
    vols = [synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=rank * n_vol + b, fields=field_names)
            for b in range(n_vol)]
    vol = vols[0]
    n_gates = vol.n_total_gates
    fields_d, masks_d = [], []
    for v in vols:
        for name in field_names:
            fields_d.append(torch.from_numpy(np.ascontiguousarray(np.ma.getdata(v.fields[name]))).to(dev))
            masks_d.append(torch.from_numpy(np.ma.getmaskarray(v.fields[name]).astype(np.uint8)).to(dev))
    shared = None
    if "RHOHV" in field_names:      # config 3: RHOHV >= 0.8 QC mask shared by all fields, evaluated on the device
        shared = rg.device_gate_mask(fields_d[field_names.index("RHOHV")], "below", 0.8)
    log(f"rank {rank}: synthetic volume(s) {cfg['n_elev']}x{cfg['n_az']}x{cfg['n_gates']} ready in {time.perf_counter() - t0:.1f}s")

    # ---- geometry (once per scan strategy; untimed) -----------------------------------------------------
    t0 = time.perf_counter()
    n_ff = n_f * n_vol
    out = torch.empty((n_ff, n_vox), dtype=torch.float32, device=dev)
    if args.mode == "csr":
        search = None
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            # single field-volume passes use the compact copy of the CSR; 'auto' keeps the reference's index array
            # next to it when both fit and builds the copy alone otherwise (config 4: 33 G pairs)
            want_compact = n_ff == 1 and not args.no_compact
            geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, tmp,
                                            layout="auto" if want_compact else "csr")
        if want_compact and geom.device_csr(dev).gate_indices is not None:
            free_b, _ = torch.cuda.mem_get_info(dev)              # room for the copy (2.3 bytes per pair + scratch)?
            want_compact = free_b > 3.2 * geom.n_pairs() + (8 << 30)
        gridder = CsrGridder(geom, n_gates, n_ff, device=dev, compact=want_compact)
        n_pairs = gridder.csr.n_pairs
        ref_format_bytes = gridder.algorithmic_bytes()          # 8 bytes per pair
        if gridder.compact is not None:
            algo_bytes = gridder.compact_bytes()                # what this kernel has to move: ~6.3 bytes per pair
            kernel_name = "csr_compact_kernel"
        else:
            algo_bytes = ref_format_bytes
            kernel_name = "csr_apply_dyn_kernel"
    else:
        from radar_processor_amd.roi_grid import roi_grid_fields_device
        search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, device=dev)
        geom = rg.GridGeometry(shape, limits, None, None, None, toa=17000.0)
        n_pairs = None
        algo_bytes = (12 + 5 * n_ff) * n_gates + 4 * n_ff * n_vox
        ref_format_bytes = algo_bytes
        kernel_name = "roi_block_kernel"
    torch.cuda.synchronize()
    t_geom = time.perf_counter() - t0
    log(f"rank {rank}: geometry ({args.mode}) ready in {t_geom:.1f}s"
        + (f": {n_pairs:,} pairs ({n_pairs / n_vox:.1f}/voxel), {algo_bytes / 1e9:.2f} GB algorithmic per launch" if n_pairs else ""))

    grid4 = out.view(n_ff, *shape)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        if args.mode == "csr":
            gridder.pack(fields_d, masks_d, shared)
            if i is not None:
                ev[i][0].record()
            gridder.apply(out)
            if i is not None:
                ev[i][1].record()
        else:
            if i is not None:
                ev[i][0].record()
            roi_grid_fields_device(search, fields_d, masks_d, shared_mask=shared, out=grid4)
            if i is not None:
                ev[i][1].record()
        for k in range(n_ff):
            rg.column_argmax(grid4[k])
            rg.constant_altitude_ppi(grid4[k], geom, 4000.0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else float("nan")

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_gpus * n_ff * n_vox / (elapsed / args.steps) / 1e6
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        workload = (f"{cfg['n_elev']}x{cfg['n_az']}x{cfg['n_gates']}-gate volume -> {shape[0]}x{shape[1]}x{shape[2]} grid, "
                    f"{'+'.join(field_names)}, {n_vol} volume(s)/GPU/step, mode={args.mode}")
        compact_on = args.mode == "csr" and gridder.compact is not None
        workload_key = f"{args.config}/{'csr_compact' if compact_on else args.mode}/F{n_f}/B{n_vol}"
        result = {
            "metric": "Mvoxels/s gridded (+ achieved HBM GB/s in roofline)",
            "value": round(value, 2),
            "unit": "Mvoxel/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "key": workload_key, "gates": n_gates, "voxels": n_vox,
                       "pairs": n_pairs, "fields_per_pass": n_ff,
                       "step": "pack_fields + " + (("csr_compact_apply" if compact_on else "csr_apply") if args.mode == "csr"
                                                   else "roi_grid")
                               + " + colmax/argmax + cappi4000 per field-volume"},
            "roofline": {
                "bound": "hbm" if args.mode == "csr" else "valu (reported against hbm)",
                "kernel": kernel_name,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic(workload_key),
                "algorithmic_bytes_per_launch": int(algo_bytes),
                "kernel_ms": round(kernel_ms, 4),
                # the same launch priced in the reference's CSR format (8 bytes per pair); larger than
                # `achieved` when the compact device copy is in use, because that kernel moves fewer bytes per pair
                "reference_format_bytes_per_launch": int(ref_format_bytes),
                "reference_format_GBps": round(ref_format_bytes / (kernel_ms * 1e-3) / 1e9, 1),
            },
        }
        if n_gpus == 1 and args.mode == "csr":
            # side measurement on the same inputs (not part of `value`): the CSR-free fused gridder (K2)
            try:
                from radar_processor_amd.roi_grid import roi_grid_fields_device
                search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, device=dev)
                roi_grid_fields_device(search, fields_d, masks_d, shared_mask=shared, out=grid4)
                a_ev, b_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_ev.record()
                for _ in range(3):
                    roi_grid_fields_device(search, fields_d, masks_d, shared_mask=shared, out=grid4)
                b_ev.record()
                b_ev.synchronize()
                k2_ms = a_ev.elapsed_time(b_ev) / 3
                result["extras"] = {"geometry_build_s": round(t_geom, 3),
                                    "roi_grid_fused_ms": round(k2_ms, 3),
                                    "roi_grid_fused_mvoxel_s": round(n_ff * n_vox / k2_ms / 1e3, 1),
                                    "note": "rg_roi_grid_f32: same volume gridded without a CSR (search fused in)"}
                del search
            except Exception as exc:
                log(f"fused side measurement failed: {exc!r}")
        if n_gpus == 1 and not args.no_cpu_baseline and args.mode == "csr":
            log("timing the CPU baseline (NumPy port, 1 core) on a bounded sample ...")
            try:
                result["cpu_baseline"] = cpu_baseline(geom, vol, field_names[0], n_vox, args.cpu_sample_pairs)
            except Exception as exc:   # never lose the GPU line to a host-side problem
                log(f"cpu baseline failed: {exc!r}")
                result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
