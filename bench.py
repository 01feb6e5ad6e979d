#!/usr/bin/env python3
"""Benchmark of the radar_grid hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config METRIC|C2|C4|C5] [--fields F] [--mode csr|fused]

One *step* = one pass of the per-volume hot path over one batch of synthetic input per GPU, inputs resident in HBM:
mask fold + field interleave (rg_pack_fields_f32) -> gridding (the dominant kernel: rg_csr_compact_apply_packed_f32 =
the row-wise kernel over the packed records of the compact CSR copy; rg_csr_compact_apply_f32 / rg_csr_apply_f32 with
--tile-kernel / --no-compact or where the weights are not codable) -> COLMAX+argmax (rg_column_reduce_f32) ->
CAPPI@4000 m (rg_cappi_lerp_f32).  The geometry (CSR) is built once on the GPU before the timed region -- that is how the
reference uses it too (once per scan strategy).

Default workload (N=1): the configuration BASELINE.json's metric is quoted on -- the 12-elevation 360x1000-gate
volume onto the 40x2000x2000 grid, one field (DBZH), one volume per GPU per step.

``--config C5`` is BASELINE config 5: a batch of ``8 x N`` seeded volumes (volume b -> rank b mod N, seed b; 64 volumes
on 8 GPUs) sharing one geometry, driven through ``batch.VolumeBatch`` exactly as a user would; a step is one pass over
this rank's 8 volumes.  Independent volumes shard with NO data-path collective (weak scaling); RCCL is used only for
the barrier / max-over-ranks of the timing.

Launching: ``python bench.py --gpus N`` starts its own N ranks (fresh child processes, one per GPU, created BEFORE
anything touches the GPU; rank 0's JSON line is passed through) unless it already runs under a launcher such as
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` (RANK / WORLD_SIZE in the environment), in
which case WORLD_SIZE must equal ``--gpus``.

Rank 0 prints ONE JSON line on stdout; progress goes to stderr.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s copy)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="METRIC",
                    help="METRIC (default), C2, C4 -- see radar_processor_amd.synthetic.CONFIGS -- or C5 (batch of 8 "
                         "volumes per GPU through batch.VolumeBatch)")
    ap.add_argument("--fields", type=int, default=1, help="fields gridded per volume in one fused CSR pass (1..3)")
    ap.add_argument("--volumes-per-gpu", type=int, default=None,
                    help="volumes per GPU per step (default 1; C5: 8)")
    ap.add_argument("--c5-grid", default="METRIC", choices=("METRIC", "C2"),
                    help="C5: grid the batch is gridded onto (METRIC = 40x2000x2000, the metric's grid)")
    ap.add_argument("--mode", choices=("csr", "fused"), default="csr",
                    help="csr = precomputed geometry + CSR kernels (K1 / K1c); fused = rg_roi_grid_f32 (K2, no CSR)")
    ap.add_argument("--products", choices=("auto", "fused", "separate"), default="auto",
                    help="C5 (and the extras.c5 side measurement): how COLMAX/argmax + CAPPI are produced -- separate = grid every "
                         "volume, then rg_column_reduce_f32 / rg_cappi_lerp_f32 on the stored grids; fused = the gridding kernel's "
                         "products epilogue (column mode, no 3-D grid in HBM: the memory-saving way, about as fast); auto = what "
                         "batch.VolumeBatch does for products=PlaneProducts (separate: measured at least as fast)")
    ap.add_argument("--settle-tries", type=int, default=4,
                    help="csr mode, packed records: placements of the record array tried during the (untimed) geometry build -- the "
                         "fastest under a 3-launch probe of the gridding kernel is kept (CsrGridder.settle_records); 1 = off. "
                         "Reported in config.records_settled")
    ap.add_argument("--c5-per-pass", type=int, default=0, choices=(0, 4, 8),
                    help="C5 / extras.c5: field-volumes one pass over the geometry fuses; 0 = what batch.VolumeBatch chooses for "
                         "the geometry (8 through the row-wise kernel where its LDS window admits eight fields, else 4)")
    ap.add_argument("--no-c5-extra", action="store_true",
                    help="skip the extras.c5 side measurement (8 seeded volumes per GPU through batch.VolumeBatch after the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-compact", action="store_true",
                    help="csr mode: run the standard 8-byte-per-pair kernel instead of the compact device copy of the "
                         "CSR (the same values to float32 rounding)")
    ap.add_argument("--layout", default="auto", choices=("auto", "csr", "compact", "packed"),
                    help="csr mode: geometry layout handed to compute_grid_geometry (auto = the reference's arrays + the "
                         "compact copy when both fit, the packed layout alone otherwise)")
    ap.add_argument("--rec-order", default="dispatch", choices=("dispatch", "segment"),
                    help="csr mode: order in which the packed records are stored -- dispatch (the order the workgroups read "
                         "them: one moving front) or segment (line-major segments, round 2's layout; A/B only)")
    ap.add_argument("--tile-kernel", action="store_true",
                    help="csr mode, A/B: run the tile kernel over the packed records (bit-identical to the standard kernel) "
                         "instead of the row-wise kernel")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend for N>1 (nccl = RCCL; gloo only for rehearsing ranks on one GPU)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --dist-backend gloo; RCCL refuses a shared GPU)")
    ap.add_argument("--pg-timeout", type=float, default=120.0,
                    help="N>1: seconds a rank waits in the rendezvous / a collective before it gives up")
    ap.add_argument("--fail-rank", type=int, default=-1,
                    help="test hook: this rank exits with code 3 before the rendezvous (exercises the launcher's failure path)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="test hook: the ranks only meet in a gloo process group (no GPU work) and rank 0 prints a JSON line")
    ap.add_argument("--cpu-sample-pairs", type=float, default=1.5e9, help="upper bound on CSR pairs in the CPU sample")
    ap.add_argument("--cpu-workers", type=int, default=0,
                    help="processes of the all-core CPU leg (0 = min(usable cores, 16): the CPU share of a 1-GPU box)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# launching N ranks from a plain `python bench.py --gpus N`
# ---------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv=None, poll_s: float = 0.2, grace_s: float = 10.0) -> int:
    """Start ``args.gpus`` fresh child processes, one per GPU, and pass rank 0's stdout through.  Nothing in THIS
    process touches the GPU (no HIP call, no exec after one): the library is compiled if missing (hipcc, CPU only),
    the device count is read with ``torch.cuda.device_count()`` (which does not initialise the runtime on this
    image), then the children are started with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set.

    Failure path: all children are polled; as soon as ONE exits non-zero its siblings -- which would otherwise sit in
    the rendezvous or in a barrier until the process-group timeout -- are terminated (SIGTERM, SIGKILL after
    ``grace_s``) and the launcher returns non-zero within seconds, without a JSON line.  Only fresh children are ever
    started; nothing that has touched the GPU is re-executed."""
    import threading
    from radar_processor_amd.build import ensure_built
    ensure_built(verbose=True)
    n = args.gpus
    if not args.share_device:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            log(f"--gpus {n} but only {have} GPU(s) are visible; refusing to label a smaller run as n_gpus={n}")
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "2")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *(sys.argv[1:] if argv is None else argv)],
                                      env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    chunks = []                       # rank 0's stdout is drained by a thread, so that polling never blocks on the pipe
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(poll_s)
    if failed is not None:
        log(f"rank {failed[0]} exited with code {failed[1]}: stopping the other ranks")
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + grace_s
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        log(f"rank exit codes: {[p.returncode for p in procs]}")
        return 1
    reader.join(timeout=30)
    # rank 0's stdout carries the ONE JSON line -- and, with the gloo rehearsal backend, a "[Gloo] Rank 0 is connected ..." line
    # that library prints there: only the JSON goes on, the rest to stderr
    for line in b"".join(chunks).decode().splitlines():
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return 0


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = NumPy restatement of radar_grid/interpolate.py:69-104), timed on the host cores
# ---------------------------------------------------------------------------------------------------------------
def _cpu_worker(job):
    """All-core leg: one process, one contiguous block of (z,y) rows of the sample, arrays memory-mapped from the
    files rank 0 wrote.  Runs in a fresh interpreter (multiprocessing 'spawn'), never touches the GPU."""
    path, r_lo, r_hi, nx = job
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    from oracle import radar_grid_oracle as oracle
    indptr = np.load(os.path.join(path, "indptr.npy"), mmap_mode="r")
    gidx = np.load(os.path.join(path, "gidx.npy"), mmap_mode="r")
    wts = np.load(os.path.join(path, "wts.npy"), mmap_mode="r")
    data = np.load(os.path.join(path, "data.npy"))
    mask = np.load(os.path.join(path, "mask.npy"))
    ip = np.asarray(indptr[r_lo * nx:r_hi * nx + 1])
    p0, p1 = int(ip[0]), int(ip[-1])
    g = np.asarray(gidx[p0:p1])
    w = np.asarray(wts[p0:p1])
    t0 = time.perf_counter()
    out = oracle.csr_apply(ip - p0, g, w, data, mask, (1, r_hi - r_lo, nx))
    dt = time.perf_counter() - t0
    return p1 - p0, dt, float(np.nansum(out[:, :1, :8]))


def _cpu_identity():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def cpu_baseline(pool, n_workers, geom, vol, field_name, n_vox_total, max_pairs):
    """Reference CPU path on a bounded sample: whole (z,y) rows around the centre of the grid, CSR rows copied from the
    GPU-built geometry.  Two legs: (a) one thread -- what radar_grid/interpolate.py does; (b) one process per usable
    core, each gridding its own block of rows of the same sample (how a host would batch independent work)."""
    import shutil
    import tempfile
    from oracle import radar_grid_oracle as oracle
    csr = geom.device_csr()
    nz, ny, nx = geom.grid_shape
    ip_rows = csr.indptr[::nx].cpu().numpy().astype(np.int64)      # pair offset at the start of every (z,y) row
    n_rows = nz * ny
    centre = (nz // 2) * ny + ny // 2
    # widest window of whole rows around the grid centre that stays under the pair budget (bisection)
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
    if csr.weights is not None:
        wts = csr.weights[p0:p1].cpu().numpy()
    else:                                          # packed-only geometry: unpack the sample's weights (lossless code)
        wts = geom.device_compact().decode_weights(csr, v0, v1).cpu().numpy()
    shape = (1, r1 - r0, nx)
    data, mask = oracle.merge_masks(vol.fields[field_name])
    oracle.csr_apply(indptr[:2], gidx[:int(indptr[1])], wts[:int(indptr[1])], data, mask, (1, 1, 1))  # touch code paths
    t0 = time.perf_counter()
    oracle.csr_apply(indptr, gidx, wts, data, mask, shape)
    dt = time.perf_counter() - t0
    pairs = p1 - p0
    pairs_per_s = pairs / dt
    total_pairs = csr.n_pairs
    model, logical, usable = _cpu_identity()
    single = {
        "value": round(n_vox_total / (total_pairs / pairs_per_s) / 1e6, 4), "unit": "Mvoxel/s", "cores": 1,
        "mpairs_per_s": round(pairs_per_s / 1e6, 2), "seconds": round(dt, 2),
        "sample_mvoxel_per_s": round((v1 - v0) / dt / 1e6, 4),
    }
    sample = (f"NumPy restatement of apply_geometry (oracle.csr_apply) on (z,y) rows [{r0},{r1}) of the bench grid's "
              f"{n_rows} around its centre: {v1 - v0} voxels, {pairs} CSR pairs; value = full-grid voxels / (all "
              f"{total_pairs} pairs / measured pair rate)")
    result = {"value": single["value"], "unit": "Mvoxel/s", "cores": 1, "kind": "port", "sample": sample,
              "cpu_model": model, "os_cpu_count": logical, "usable_cores": usable, "single_thread": single,
              "all_cores": None}
    if pool is None:
        return result
    # ---- all-core leg: the same sample, split into one block of rows per worker (equal pair counts) -------------
    workers = n_workers
    tmp = tempfile.mkdtemp(prefix="rg_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        np.save(os.path.join(tmp, "indptr.npy"), indptr)
        np.save(os.path.join(tmp, "gidx.npy"), gidx)
        np.save(os.path.join(tmp, "wts.npy"), wts)
        np.save(os.path.join(tmp, "data.npy"), np.ascontiguousarray(data))
        np.save(os.path.join(tmp, "mask.npy"), np.ascontiguousarray(mask))
        del gidx, wts
        row_pairs = indptr[::nx]
        cuts = np.searchsorted(row_pairs, np.linspace(0, pairs, workers + 1)[1:-1])
        edges = [0, *[int(c) for c in cuts], r1 - r0]
        jobs = [(tmp, edges[i], edges[i + 1], nx) for i in range(workers) if edges[i + 1] > edges[i]]
        pool.map(_cpu_worker, [(tmp, 0, 1, nx)] * workers)            # import NumPy / the oracle in every worker
        best = None
        for _ in range(2):       # all workers run concurrently; the leg takes as long as its slowest worker computes
            done = pool.map(_cpu_worker, jobs, chunksize=1)
            slowest = max(d[1] for d in done)
            best = slowest if best is None else min(best, slowest)
        rate = sum(d[0] for d in done) / best
        result["all_cores"] = {
            "value": round(n_vox_total / (total_pairs / rate) / 1e6, 4), "unit": "Mvoxel/s", "cores": len(jobs),
            "mpairs_per_s": round(rate / 1e6, 2), "seconds": round(best, 2),
            "how": "one process per core of the box's CPU share (multiprocessing spawn), NumPy single-threaded, each gridding its own "
                   "block of whole rows of the sample (equal pair counts), all at once; compute time of the slowest worker, best of 2",
        }
        result["value"], result["cores"] = result["all_cores"]["value"], len(jobs)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return result


def pmc_traffic(workload: str):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (profiles/pmc_traffic.json, committed with
    the rocprof CSVs it was derived from); (None, reason) when no measurement exists for this workload."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
        rec = table.get(workload)
        if rec:
            return rec.get("hbm_bytes_per_launch"), (f"profiles/pmc_traffic.json[{workload!r}] -- rocprofv3 --pmc FETCH_SIZE / "
                                                      f"WRITE_SIZE passes of this command ({rec.get('tag', 'tag n/a')}), "
                                                      "not collected during this run")
    except Exception:
        pass
    return None, "no committed PMC pass for this workload"


def measured_read_ceiling(torch, rg, dev, buffers):
    """Read bandwidth a pure streaming kernel gets on THIS box right now (rg_stream_read_probe over the largest
    resident buffer, best of 5 after a warm-up).  Untimed side measurement, outside the timed region."""
    from radar_processor_amd import _native
    lib = rg.load_library()
    buf = max((b for b in buffers if b is not None and b.numel() > 0), key=lambda b: b.numel() * b.element_size(),
              default=None)
    if buf is None:
        return None
    nbytes = (buf.numel() * buf.element_size()) // 16 * 16
    if nbytes < (1 << 28):                       # a small geometry would measure the Infinity Cache, not HBM
        scratch = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        buf, nbytes = scratch, scratch.numel()
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    best = None
    for i in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _native.check(lib.rg_stream_read_probe(_native.ptr(buf), nbytes, _native.ptr(sink), _native.stream_ptr()),
                      "rg_stream_read_probe")
        e1.record()
        e1.synchronize()
        if i:
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
    return nbytes / (best * 1e-3) / 1e9


def gather_per_rank(dist, backend, dev, value: float):
    """One float from every rank, in rank order (all ranks call this; every rank gets the list)."""
    import torch
    world = dist.get_world_size()
    t = torch.tensor([value], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return [round(float(x.item()), 4) for x in parts]


def c5_side_measurement(args, rg, batch, synthetic, torch, dist, dev, geom_or_search, grid_geom, cfg_gates, rank, world, steps=3):
    """north_star's batch figure next to ANY bench line: 8 seeded volumes per GPU (volume b -> rank b mod N, seed b; 64 volumes
    on 8 GPUs) sharing one geometry, through batch.VolumeBatch, products = COLMAX + argmax + CAPPI@4000 m per volume;
    barrier-bracketed, max over ranks.  Both ways of producing the planes are timed (separate kernels on stored grids / the
    epilogue of the gridding kernel).  Collective-safe: a rank whose part fails keeps meeting the others in every barrier and
    all-reduce, and the failure is reported in the result instead of hanging or killing the bench line."""
    from radar_processor_amd.gridding import PlaneProducts
    per_gpu = 8
    total = per_gpu * world
    err = None
    vols, vb = [None] * total, None
    try:
        for b in batch.shard_indices(total, rank, world):
            v = synthetic.make_volume(cfg_gates["n_elev"], cfg_gates["n_az"], cfg_gates["n_gates"], seed=b, fields=("DBZH",))
            vols[b] = {"DBZH": (torch.from_numpy(np.ascontiguousarray(np.ma.getdata(v.fields["DBZH"]))).to(dev),
                                torch.from_numpy(np.ma.getmaskarray(v.fields["DBZH"]).astype(np.uint8)).to(dev))}
        vb = batch.VolumeBatch(geom_or_search, ("DBZH",), device=dev)
        if args.c5_per_pass and args.mode == "csr":
            vb._cap = args.c5_per_pass
    except Exception as exc:
        err = repr(exc)

    def separate(g):
        return [(rg.column_argmax(g[k]), rg.constant_altitude_ppi(g[k], grid_geom, 4000.0)) for k in range(g.shape[0])]
    modes = {"separate": separate, "fused": PlaneProducts(colmax=True, argmax=True, cappi=(4000.0,), fused=True)}
    out = {}

    def meet():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for name, products in modes.items():
        if name == "fused" and args.mode != "csr":
            continue                                    # the CSR-free gridder has no epilogue
        events = []

        def one(timed=False):
            return vb.grid_shard(vols, products=products, rank=rank, world_size=world, events=events if timed else None)
        try:
            if err is None:
                one()
        except Exception as exc:
            err = repr(exc)
        meet()
        t0 = time.perf_counter()
        try:
            if err is None:
                for _ in range(steps):
                    one(True)
        except Exception as exc:
            err = repr(exc)
        meet()
        dt = (time.perf_counter() - t0) / steps
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        pass_ms = [a.elapsed_time(b) for a, b in events] if err is None else []
        out[name] = {"ms_per_step": round(dt * 1e3, 3), "pass_ms_median": round(float(np.median(pass_ms)), 3) if pass_ms else None,
                     "passes_per_step": len(pass_ms) / steps,
                     "volumes_per_pass": (vb.volumes_per_pass if vb is not None and err is None else None)}
    bad = 0.0 if err is None else 1.0
    if world > 1:
        t = torch.tensor([bad], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bad = float(t.item())
    if bad:
        return {"failed": err or "another rank failed"}, total
    return out, total


# ---------------------------------------------------------------------------------------------------------------
def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"WORLD_SIZE={world} but --gpus {args.gpus}: start this script as `python bench.py --gpus {args.gpus}` (it "
            f"spawns its own ranks) or under `python -m torch.distributed.run --nproc-per-node {args.gpus}`")
        return 2

    if args.fail_rank == rank:
        log(f"rank {rank}: --fail-rank, exiting before the rendezvous")
        return 3

    if args.rendezvous_only:                 # launcher rehearsal on a box without GPUs: rendezvous, barrier, done
        from datetime import timedelta
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", timeout=timedelta(seconds=args.pg_timeout))
        dist.barrier()
        per_rank = gather_per_rank(dist, "gloo", None, 10.0 + rank)        # the same gather the real run uses
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "n_gpus": dist.get_world_size(), "per_rank_kernel_ms": per_rank}), flush=True)
        dist.destroy_process_group()
        return 0

    # ---- everything that creates processes happens BEFORE the GPU is touched ---------------------------------
    from radar_processor_amd.build import ensure_built
    ensure_built(verbose=rank == 0)              # bare checkout: compile the git-ignored library (file-locked)
    pool, n_workers = None, 1
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "csr"
    if want_cpu:
        import multiprocessing as mp
        _, _, usable = _cpu_identity()
        # a 1-GPU lease of this pool comes with a CPU share of 16 cores, whatever the affinity mask says (256 on the
        # EPYC 9575F hosts): one worker per core of that share unless --cpu-workers says otherwise
        n_workers = args.cpu_workers or min(usable, 16)
        if n_workers > 1:
            pool = mp.get_context("spawn").Pool(n_workers)

    import torch
    import torch.distributed as dist
    import radar_processor_amd as rg
    from radar_processor_amd import batch, synthetic
    from radar_processor_amd.gridding import CsrGridder

    n_gpus = world
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from datetime import timedelta
        pg_timeout = timedelta(seconds=args.pg_timeout)       # a rank that never arrives fails the others, not hangs them
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, timeout=pg_timeout)      # RCCL over xGMI
        else:
            dist.init_process_group(backend="gloo", timeout=pg_timeout)
        assert dist.get_world_size() == world
    rg.load_library()

    c5 = args.config == "C5"
    cfg = dict(synthetic.CONFIGS["C2" if c5 else args.config])
    if c5:
        cfg["grid_shape"] = synthetic.CONFIGS[args.c5_grid]["grid_shape"]
        cfg["grid_limits"] = synthetic.CONFIGS[args.c5_grid]["grid_limits"]
    field_names = ("DBZH", "ZDR", "RHOHV")[:max(1, min(3, args.fields))]
    n_f = len(field_names)
    n_vol = max(1, args.volumes_per_gpu if args.volumes_per_gpu is not None else (8 if c5 else 1))
    if not c5 and n_f * n_vol > 8:
        raise SystemExit("fields x volumes-per-gpu must be <= 8 (one fused pass)")
    shape, limits = cfg["grid_shape"], cfg["grid_limits"]
    n_vox = int(np.prod(shape))

    # ---- synthetic inputs, resident in HBM before the timed region ------------------------------------
    t0 = time.perf_counter()
    total_vol = n_vol * world
    my_vols = batch.shard_indices(total_vol, rank, world)           # volume b -> rank b mod N; seed = b
    vols = {b: synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=b, fields=field_names)
            for b in my_vols}
    vol = vols[my_vols[0]]
    n_gates = vol.n_total_gates
    fields_d, masks_d, dev_volumes = [], [], [None] * total_vol
    for b in my_vols:
        entry = {}
        for name in field_names:
            f_t = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(vols[b].fields[name]))).to(dev)
            m_t = torch.from_numpy(np.ma.getmaskarray(vols[b].fields[name]).astype(np.uint8)).to(dev)
            fields_d.append(f_t)
            masks_d.append(m_t)
            entry[name] = (f_t, m_t)
        dev_volumes[b] = entry
    shared = None
    if "RHOHV" in field_names and not c5:   # config 3: RHOHV >= 0.8 QC mask shared by all fields, on the device
        shared = rg.device_gate_mask(fields_d[field_names.index("RHOHV")], "below", 0.8)
    log(f"rank {rank}: {len(my_vols)} synthetic volume(s) {cfg['n_elev']}x{cfg['n_az']}x{cfg['n_gates']} ready in "
        f"{time.perf_counter() - t0:.1f}s")

    # ---- geometry (once per scan strategy; untimed) -----------------------------------------------------
    t0 = time.perf_counter()
    n_ff = n_f * n_vol
    gridder = search = None
    if args.mode == "csr":
        import tempfile
        from radar_processor_amd import _native, grid_geometry
        grid_geometry.DEFAULT_REC_ORDER = (_native.RG_REC_ORDER_DISPATCH if args.rec_order == "dispatch"
                                           else _native.RG_REC_ORDER_SEGMENT)
        with tempfile.TemporaryDirectory() as tmp:
            # passes run through the compact copy of the CSR; 'auto' keeps the reference's index array next to it when
            # both fit and builds the copy alone otherwise (config 4: 33 G pairs)
            want_compact = not args.no_compact
            geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, tmp,
                                            layout=args.layout if want_compact else "csr")
            # C5: VolumeBatch fuses 8 or 4 field-volumes into one pass (gridding.volumes_per_pass_cap: per geometry)
            from radar_processor_amd.gridding import volumes_per_pass_cap
            c5_cap = args.c5_per_pass or volumes_per_pass_cap(geom, dev)
            fields_per_pass = min(c5_cap if c5 else 8, n_ff)
        if want_compact and geom.device_csr(dev).gate_indices is not None:
            free_b, _ = torch.cuda.mem_get_info(dev)              # room for the copy (2.3 bytes per pair + scratch)?
            want_compact = free_b > 3.2 * geom.n_pairs() + (8 << 30)
        gridder = CsrGridder(geom, n_gates, fields_per_pass, device=dev, compact=want_compact)
        n_pairs = gridder.csr.n_pairs
        settled = settle_out = None
        torch.cuda.synchronize()
        t_built = time.perf_counter() - t0                       # the geometry itself; what follows is optional tuning
        if gridder.compact is not None and gridder.packed_stream and args.settle_tries > 1 and not args.tile_kernel:
            gridder.pack(fields_d[:fields_per_pass] if len(fields_d) >= fields_per_pass else (fields_d * fields_per_pass)[:fields_per_pass],
                         (masks_d[:fields_per_pass] if len(masks_d) >= fields_per_pass else (masks_d * fields_per_pass)[:fields_per_pass]))
            # probed on the very grid the timed steps will write (the draw is about where records AND grid lie)
            settle_out = None if c5 else torch.empty((n_ff, n_vox), dtype=torch.float32, device=dev)
            settled = gridder.settle_records(tries=args.settle_tries, out=settle_out)   # part of the one-off geometry build: untimed
            if settled is not None:
                torch.cuda.synchronize()
                settled["seconds"] = round(time.perf_counter() - t0 - t_built, 3)
        ref_format_bytes = gridder.algorithmic_bytes()          # 8 bytes per pair
        if gridder.compact is not None:
            algo_bytes = gridder.compact_bytes()                # what this kernel has to move: ~5.7 bytes per pair
            if gridder.packed_stream and args.tile_kernel:
                gridder.tile = 384                              # A/B: the tile kernel over the same packed records
            kernel_name = ("csr_compact_rowwise_kernel" if gridder.packed_stream and gridder.tile == 0
                           else "csr_compact_kernel")
        else:
            algo_bytes = ref_format_bytes
            kernel_name = "csr_apply_dyn_kernel"
        batch_geometry = geom
    else:
        from radar_processor_amd.roi_grid import roi_grid_fields_device
        search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, shape, limits, device=dev)
        geom = rg.GridGeometry(shape, limits, None, None, None, toa=17000.0)
        n_pairs = None
        fields_per_pass = min(8, n_ff)
        algo_bytes = (12 + 5 * fields_per_pass) * n_gates + 4 * fields_per_pass * n_vox
        ref_format_bytes = algo_bytes
        kernel_name = "roi_block_kernel"
        batch_geometry = search
    torch.cuda.synchronize()
    t_geom = time.perf_counter() - t0
    if args.mode == "csr":
        t_geom = t_built                             # extras.geometry_build_s: the build; the settling reports its own seconds
    log(f"rank {rank}: geometry ({args.mode}) ready in {t_geom:.1f}s"
        + (f": {n_pairs:,} pairs ({n_pairs / n_vox:.1f}/voxel), {algo_bytes / 1e9:.2f} GB algorithmic per launch" if n_pairs else ""))

    events = []                                   # (start, end) per gridding launch inside the timed region
    kernel_mode = None

    if c5:
        vb = batch.VolumeBatch(batch_geometry, field_names, device=dev)
        if args.c5_per_pass and args.mode == "csr":
            vb._cap = args.c5_per_pass
        from radar_processor_amd.gridding import PlaneProducts

        def reducer(g):
            return [(rg.column_argmax(g[k]), rg.constant_altitude_ppi(g[k], geom, 4000.0)) for k in range(g.shape[0])]
        # the same planes either way (tests/test_gpu_columns.py, test_gpu_batch.py): COLMAX + argmax + CAPPI@4000 m per volume
        c5_products = (reducer if (args.products == "separate" or args.mode != "csr")
                       else PlaneProducts(cappi=(4000.0,), fused=True if args.products == "fused" else None))

        if args.mode == "csr" and isinstance(c5_products, PlaneProducts) and gridder.has_columns_kernel:
            if c5_products.fused:                        # (--products fused) the passes run the epilogue: no 3-D store
                algo_bytes = gridder.columns_bytes(store_grid=False, n_keep=2, colmax=True)
                ref_format_bytes -= fields_per_pass * 4 * n_vox
                kernel_mode = "column mode, products epilogue (no 3-D store)"

        def step(timed=False):
            return vb.grid_shard(dev_volumes, products=c5_products, rank=rank, world_size=world,
                                 events=events if timed else None)
    else:
        out = settle_out if (args.mode == "csr" and settle_out is not None) else torch.empty((n_ff, n_vox), dtype=torch.float32, device=dev)
        grid4 = out.view(n_ff, *shape)

        def step(timed=False):
            if timed:
                pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            if args.mode == "csr":
                gridder.pack(fields_d, masks_d, shared)
                if timed:
                    pair[0].record()
                gridder.apply(out)
            else:
                if timed:
                    pair[0].record()
                roi_grid_fields_device(search, fields_d, masks_d, shared_mask=shared, out=grid4)
            if timed:
                pair[1].record()
                events.append(pair)
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
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms_per_step = [round(elapsed / args.steps * 1e3, 4)]
    if world > 1:
        # every rank's own wall time for the K steps (between the same two barriers), then the max the contract asks for
        per_rank_ms_per_step = gather_per_rank(dist, args.dist_backend, dev, elapsed / args.steps * 1e3)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    launch_ms = np.array([a.elapsed_time(b) for a, b in events], dtype=np.float64)
    kernel_ms = float(launch_ms.mean()) if events else float("nan")
    # one stderr line per rank: an N-GPU line (max over ranks) can be attributed to the rank that set it
    if events:
        log(f"rank {rank}: {elapsed / args.steps * 1e3:.3f} ms/step, gridding launches min/median/mean/max "
            f"{launch_ms.min():.3f}/{np.median(launch_ms):.3f}/{kernel_ms:.3f}/{launch_ms.max():.3f} ms over {len(events)}")

    kernel_ms_median_rank = float(np.median(launch_ms)) if events else float("nan")
    per_rank_kernel_ms = ([round(kernel_ms_median_rank, 4)] if world == 1
                          else gather_per_rank(dist, args.dist_backend, dev, kernel_ms_median_rank))
    # ---- north_star's 64-volume figure next to every line: 8 seeded volumes per GPU through VolumeBatch, after the timed
    # region, on every rank (max over ranks) ----------------------------------------------------------------------------
    c5_extra = None
    if not c5 and not args.no_c5_extra and args.config in ("METRIC", "C2"):
        try:
            torch.cuda.empty_cache()
            c5_extra, c5_total = c5_side_measurement(args, rg, batch, synthetic, torch, dist, dev, batch_geometry, geom,
                                                     synthetic.CONFIGS["C2"], rank, world)
            c5_extra = {"volumes_total": c5_total, "volumes_per_gpu": 8, "grid": list(shape), "mode": args.mode, **c5_extra,
                        "what": "8 seeded volumes per GPU (volume b -> rank b mod N) through batch.VolumeBatch: gridding + COLMAX/"
                                "argmax + CAPPI@4000 m per volume, inputs resident; barrier-bracketed, max over ranks; 'separate' = "
                                "grids stored, rg_column_reduce_f32 + rg_cappi_lerp_f32 on them (the default); 'fused' = the products "
                                "epilogue of the gridding kernel (column mode: no 3-D grid in HBM)"}
            for k in ("separate", "fused"):
                if k in c5_extra and isinstance(c5_extra[k], dict):
                    c5_extra[k]["mvoxel_s_all_gpus"] = round(c5_total * n_vox / (c5_extra[k]["ms_per_step"] * 1e-3) / 1e6, 1)
        except Exception as exc:                   # (the measurement itself is collective-safe; this is the bookkeeping)
            log(f"rank {rank}: extras.c5 failed: {exc!r}")
            c5_extra = {"failed": repr(exc)}

    # ---- the grid that was just timed is checked, outside the timed region (rank 0) ------------------------------
    # the compact kernel against the reference-format kernel on the same inputs, every voxel: the row-wise kernel to
    # north_star's float32 tolerance (it sums in another order), the tile kernel over the same records bit for bit; and
    # the filled / empty pattern against the row pointers.  (Parity against the oracle: tests/test_gpu_fullsize.py.)
    checked = None
    if rank == 0 and not c5 and args.mode == "csr":
        csr_now = gridder.csr
        nonempty = (csr_now.indptr[1:] - csr_now.indptr[:-1]) > 0
        filled = torch.isfinite(out[0])
        assert bool((filled <= nonempty).all()), "a voxel without neighbours was filled"
        checked = f"{int(filled.sum())} of {n_vox} voxels filled, none of them in an empty row"
        if gridder.compact is not None and csr_now.gate_indices is not None:
            ref_gridder = CsrGridder(geom, n_gates, fields_per_pass, device=dev, compact=False)
            ref_gridder.pack(fields_d, masks_d, shared)
            ref_out = torch.empty_like(out)
            ref_gridder.apply(ref_out)
            if kernel_name == "csr_compact_rowwise_kernel":
                assert bool(torch.equal(torch.isnan(ref_out), torch.isnan(out))), "filled voxels differ from rg_csr_apply_f32"
                scale = max(float(f_t[torch.isfinite(f_t)].abs().max()) for f_t in fields_d)
                # |diff| <= 1e-5*|ref| + 2e-6*max|field|: north_star's relative bar plus an absolute floor three times the
                # largest absolute error ever measured (6.1e-7*max|field|; tests/conftest.py: ATOL_FRAC)
                excess = torch.nan_to_num((out - ref_out).abs() - 1e-5 * ref_out.abs(), nan=0.0)
                worst = float(excess.max())
                assert worst <= 2e-6 * scale, "row-wise kernel and reference-format kernel differ beyond 1e-5 rel + 2e-6 abs"
                sig = ref_out.abs() > 1e-3 * scale
                rel = float(((out - ref_out).abs()[sig] / ref_out.abs()[sig]).max())
                assert rel <= 1e-5, "row-wise kernel beyond 1e-5 relative on significant voxels"
                checked += (f"; row-wise kernel vs rg_csr_apply_f32 on all of them: same voxels filled, |diff| <= 1e-5*|ref| + "
                            f"2e-6*max|field| everywhere (worst relative deviation where |ref| > 1e-3*max|field|: {rel:.1e})")
                del excess, sig
                gridder.tile = 384                       # and the tile kernel over the very same records: bit for bit
                gridder.apply(out)
                gridder.tile = 0
                assert bool(torch.equal(ref_out.view(torch.int32), out.view(torch.int32))), \
                    "tile kernel over the packed records and reference-format kernel disagree"
                checked += "; tile kernel over the same packed records == rg_csr_apply_f32 bit for bit"
            else:
                assert bool(torch.equal(ref_out.view(torch.int32), out.view(torch.int32))), \
                    "compact kernel and reference-format kernel disagree on the timed grid"
                checked += "; compact tile kernel == rg_csr_apply_f32 bit for bit on all of them"
            del ref_out, ref_gridder
        elif gridder.compact is not None and kernel_name == "csr_compact_rowwise_kernel":
            # Packed-only geometry (layout 'auto' from 50 M pairs on): the reference's index / weight arrays do not exist.
            # (1) the TILE kernel over the same records (rg_csr_apply_f32's order, its bits on every geometry the tests
            # compare them on) against the row-wise grid on every voxel; (2) bands of whole (z, y) rows -- the grid's first
            # rows, its middle, its last -- DECODED from the records back into the reference's CSR (CompactCSR.decode /
            # decode_weights: bit-exact) and gridded by rg_csr_apply_f32 itself as a stand-alone geometry, against both.
            from radar_processor_amd.grid_geometry import DeviceCSR, GridGeometry
            scale = max(float(f_t[torch.isfinite(f_t)].abs().max()) for f_t in fields_d)
            tile_out = torch.empty_like(out)
            gridder.tile = 384
            gridder.apply(tile_out)
            gridder.tile = 0
            assert bool(torch.equal(torch.isnan(tile_out), torch.isnan(out))), "filled voxels differ between the two kernels"
            excess = torch.nan_to_num((out - tile_out).abs() - 1e-5 * tile_out.abs(), nan=0.0)
            assert float(excess.max()) <= 2e-6 * scale, "row-wise and tile kernel differ beyond 1e-5 rel + 2e-6 abs"
            sig = tile_out.abs() > 1e-3 * scale
            rel = float(((out - tile_out).abs()[sig] / tile_out.abs()[sig]).max())
            assert rel <= 1e-5, "row-wise kernel beyond 1e-5 relative on significant voxels"
            del excess, sig
            nz_, ny_, nx_ = shape
            band = 24                                         # (z, y) rows per band: ~50 k voxels, a few M pairs each
            rows_total = nz_ * ny_
            bands, decoded_pairs = [0, (nz_ // 2) * ny_ + ny_ // 2 - band // 2, rows_total - band], 0
            for r_lo in bands:
                v0, v1 = r_lo * nx_, (r_lo + band) * nx_
                ip = csr_now.indptr[v0:v1 + 1].to(torch.int64)
                sub = DeviceCSR((ip - ip[0]).to(torch.int32), gridder.compact.decode(csr_now, v0, v1),
                                gridder.compact.decode_weights(csr_now, v0, v1).clone(), csr_now.max_gate)
                decoded_pairs += sub.n_pairs
                sub_geom = GridGeometry.from_device((1, band, nx_), ((0.0, 0.0), limits[1], limits[2]), sub, geom.toa)
                k1 = CsrGridder(sub_geom, n_gates, fields_per_pass, device=dev, compact=False)
                k1.packed = gridder.packed                    # the fields as packed for the timed pass
                k1_out = torch.empty((n_ff, band * nx_), dtype=torch.float32, device=dev)
                k1.apply(k1_out)
                assert bool(torch.equal(k1_out.view(torch.int32), tile_out[:, v0:v1].contiguous().view(torch.int32))), \
                    "tile kernel over the packed records and rg_csr_apply_f32 on the decoded rows disagree"
                del sub, sub_geom, k1, k1_out
            checked += (f"; packed-only geometry: row-wise kernel vs the tile kernel over the same records on all of them (same "
                        f"voxels filled, |diff| <= 1e-5*|ref| + 2e-6*max|field|, worst relative deviation where |ref| > "
                        f"1e-3*max|field|: {rel:.1e}); tile kernel == rg_csr_apply_f32 bit for bit on {len(bands)} bands of {band} "
                        f"whole (z,y) rows decoded back into the reference's CSR ({decoded_pairs} pairs)")
            del tile_out
        del nonempty, filled
    launches_per_step = len(events) / max(args.steps, 1)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_gpus * n_ff * n_vox / (elapsed / args.steps) / 1e6
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        workload = (f"{cfg['n_elev']}x{cfg['n_az']}x{cfg['n_gates']}-gate volume -> {shape[0]}x{shape[1]}x{shape[2]} grid, "
                    f"{'+'.join(field_names)}, {n_vol} volume(s)/GPU/step"
                    + (f" (BASELINE config 5: {total_vol} seeded volumes, volume b -> rank b mod {world}, through "
                       f"batch.VolumeBatch)" if c5 else "") + f", mode={args.mode}")
        compact_on = args.mode == "csr" and gridder.compact is not None
        rowwise_on = compact_on and kernel_name == "csr_compact_rowwise_kernel"
        workload_key = (f"{args.config}/{'csr_rowwise' if rowwise_on else 'csr_compact' if compact_on else args.mode}"
                        f"{'_products' if kernel_mode else ''}/F{n_f}/B{n_vol}")
        traffic, traffic_source = pmc_traffic(workload_key)
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
            "per_rank_kernel_ms": per_rank_kernel_ms,
            "per_rank_ms_per_step": per_rank_ms_per_step,
            "config": {"workload": workload, "key": workload_key, "gates": n_gates, "voxels": n_vox,
                       "pairs": n_pairs, "fields_per_pass": fields_per_pass, "volumes_total": total_vol,
                       "ranks_seen_by_process_group": dist.get_world_size() if world > 1 else 1,
                       "checked": checked,
                       "products": (args.products if c5 else None),
                       "records_settled": (settled if args.mode == "csr" else None),
                       "record_order": (args.rec_order if args.mode == "csr" and gridder.compact is not None
                                 and gridder.packed_stream else None),
                "step": "pack_fields + " + (("csr_compact_apply" if compact_on else "csr_apply") if args.mode == "csr"
                                                   else "roi_grid")
                               + " + colmax/argmax + cappi4000 per field-volume"},
            "roofline": {
                "bound": "hbm" if args.mode == "csr" else "valu (reported against hbm)",
                "kernel": kernel_name,
                "kernel_mode": kernel_mode,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": int(algo_bytes),
                "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_min": round(float(launch_ms.min()), 4) if events else None,
                "kernel_ms_median": round(float(np.median(launch_ms)), 4) if events else None,
                "kernel_ms_max": round(float(launch_ms.max()), 4) if events else None,
                "launches_timed": len(events),
                "launches_per_step": launches_per_step,
                # the same launch priced in the reference's CSR format (8 bytes per pair); larger than
                # `achieved` when the compact device copy is in use, because that kernel moves fewer bytes per pair
                "reference_format_bytes_per_launch": int(ref_format_bytes),
                "reference_format_GBps": round(ref_format_bytes / (kernel_ms * 1e-3) / 1e9, 1),
                "reference_format_frac": round(ref_format_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "bytes_note": ("achieved/frac count the bytes THIS kernel has to move (its own compact format); the "
                               "reference_format_* figures price the same launch in SURVEY 8(d)'s 8-bytes-per-pair CSR and "
                               "can exceed the HBM peak: they measure throughput, not bandwidth"),
            },
        }
        try:        # what a pure streaming read gets on this box, same process, outside the timed region
            csr_now = gridder.csr if gridder is not None else None
            comp_now = gridder.compact if gridder is not None else None
            ceiling = measured_read_ceiling(torch, rg, dev, [csr_now.weights if csr_now is not None else None,
                                                             comp_now.rec if comp_now is not None else None,
                                                             search.sorted_gates if search is not None else None])
            if ceiling:
                result["roofline"]["ceiling_measured"] = round(ceiling, 1)
                result["roofline"]["frac_of_ceiling"] = round(achieved / ceiling, 4)
                result["roofline"]["ceiling_how"] = ("rg_stream_read_probe: one dwordx4 per lane over the largest resident "
                                                     "buffer, workgroups in address order, no stores, best of 5, same "
                                                     "process, untimed (a pure read; the gridding kernel also writes the grid)")
        except Exception as exc:
            log(f"ceiling probe failed: {exc!r}")
        if c5:
            # end-to-end leg: the same shard with the fields uploaded from page-locked host memory and the
            # 2-D product planes downloaded inside the timed region -- reported next to `value`, never as `value`
            try:
                host_vols = [None] * total_vol
                for b in my_vols:
                    host_vols[b] = {k: (v[0].cpu().pin_memory(), v[1].cpu().pin_memory()) for k, v in dev_volumes[b].items()}

                # page-locked landing buffers, allocated once; the downloads run on their own stream behind an event, so
                # that they overlap the next pass (VolumeBatch uploads one group ahead on a third stream), and the step
                # synchronises the device once at its end
                ny_, nx_ = shape[1], shape[2]
                pinned = {}
                d2h = torch.cuda.Stream()

                def to_host(g):
                    # the planes of this pass are parked in persistent device buffers on the compute stream (48 MB of
                    # device-to-device copies per volume), then downloaded from there on the d2h stream: the grids
                    # themselves never have to outlive the pass
                    res = []
                    for k, (a, c) in enumerate(reducer(g)):
                        if (to_host.calls, k) not in pinned:
                            pinned[(to_host.calls, k)] = (
                                [torch.empty((ny_, nx_), dtype=dt).pin_memory() for dt in (torch.float32, torch.int32, torch.float32)],
                                [torch.empty((ny_, nx_), dtype=dt, device=dev) for dt in (torch.float32, torch.int32, torch.float32)])
                        host_b, dev_b = pinned[(to_host.calls, k)]
                        for dst, src in zip(dev_b, (a[0], a[1], c)):
                            dst.copy_(src)
                        res.append(host_b)
                    d2h.wait_event(torch.cuda.current_stream().record_event())
                    with torch.cuda.stream(d2h):
                        for k in range(len(res)):
                            host_b, dev_b = pinned[(to_host.calls, k)]
                            for dst, src in zip(host_b, dev_b):
                                dst.copy_(src, non_blocking=True)
                    to_host.calls += 1
                    return res
                to_host.calls = 0

                def e2e_step():
                    to_host.calls = 0
                    r = vb.grid_shard(host_vols, products=to_host, rank=rank, world_size=world)
                    torch.cuda.synchronize()
                    return r
                e2e_step()
                t1 = time.perf_counter()
                reps = 3
                for _ in range(reps):
                    e2e_step()
                e2e = (time.perf_counter() - t1) / reps
                result["end_to_end"] = {"ms_per_step": round(e2e * 1e3, 3),
                                        "mvoxel_s_this_rank": round(n_ff * n_vox / e2e / 1e6, 1),
                                        "what": "H2D of fields+masks from page-locked host memory (one group of 4 volumes ahead, copy "
                                                "stream), gridding, COLMAX/argmax/CAPPI, D2H of the 2-D planes into page-locked "
                                                "buffers (own stream); rank 0 only"}
            except Exception as exc:
                log(f"end-to-end leg failed: {exc!r}")
        resident = None
        if gridder is not None:
            resident = gridder.csr.nbytes() + (gridder.compact.nbytes() if gridder.compact is not None else 0)
        elif search is not None:
            resident = search.sorted_gates.numel() * 4 + search.cell_start.numel() * 4
        result["extras"] = {"geometry_build_s": round(t_geom, 3),
                            "geometry_resident_gb": round(resident / 1e9, 2) if resident is not None else None,
                            "geometry_layout": (None if gridder is None else "packed" if gridder.csr.weights is None else
                                                "compact" if gridder.csr.gate_indices is None else "csr+compact"
                                                if gridder.compact is not None else "csr")}
        if c5_extra is not None:
            result["extras"]["c5"] = c5_extra
        if n_gpus == 1 and args.mode == "csr" and not c5:
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
                result.setdefault("extras", {}).update({"geometry_build_s": round(t_geom, 3),
                                    "roi_grid_fused_ms": round(k2_ms, 3),
                                    "roi_grid_fused_mvoxel_s": round(n_ff * n_vox / k2_ms / 1e3, 1),
                                    "note": "rg_roi_grid_f32: same volume gridded without a CSR (search fused in)"})
                del search
            except Exception as exc:
                log(f"fused side measurement failed: {exc!r}")
        if want_cpu:
            log("timing the CPU baseline (NumPy port: 1 thread, then all usable cores) on a bounded sample ...")
            try:
                result["cpu_baseline"] = cpu_baseline(pool, n_workers, geom, vol, field_names[0], n_vox, args.cpu_sample_pairs)
            except Exception as exc:   # never lose the GPU line to a host-side problem
                log(f"cpu baseline failed: {exc!r}")
                result["cpu_baseline"] = None
        elif world > 1:
            result["cpu_baseline"] = {"skipped": "N>1: the host-core baseline is timed on rank 0 of the N=1 line only"}
        print(json.dumps(result), flush=True)
    if pool is not None:
        pool.close()
        pool.join()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args)             # the parent never touches the GPU
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
