#!/usr/bin/env python3
"""Experiment: cost of the K1 field gather as a function of which 64 pairs share a wave-instruction.
  A  lane-contiguous pairs (today's mapping): 64 consecutive pairs of the CSR;
  B  row-mapped: 16 consecutive rows x 4 consecutive elements of each (k-th neighbours of adjacent voxels together);
each with the original gate numbering and with gates renumbered in the builder's cell-sorted order."""
import ctypes
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def row_mapped_order(indptr, rows_per_group=16, lanes_per_row=4):
    """Permutation of pair positions: groups of 16 rows, step s takes elements 4s..4s+3 of every row (rows shorter
    than the group's longest repeat their last element, as idle lanes would re-read a cached line)."""
    n_rows = len(indptr) - 1
    out = []
    for g0 in range(0, n_rows, rows_per_group):
        st = indptr[g0:g0 + rows_per_group]
        ln = indptr[g0 + 1:g0 + rows_per_group + 1] - st
        if ln.max(initial=0) == 0:
            continue
        steps = int(-(-ln.max() // lanes_per_row))
        k = (np.arange(steps)[:, None, None] * lanes_per_row + np.arange(lanes_per_row)[None, None, :])   # [s,1,j]
        k = np.minimum(k, np.maximum(ln, 1)[None, :, None] - 1)
        pos = st[None, :, None] + k
        pos = pos[:, ln > 0, :]
        out.append(pos.reshape(-1))
    return np.concatenate(out)


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    lib = ctypes.CDLL(os.path.join(HERE, "libgather_probe.so"))
    lib.gp_gather_sum.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.gp_stream_only.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    cfg = synthetic.CONFIGS["C2"]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"])
    csr = search.build_csr("barnes2")
    dev = search.dev
    nz, ny, nx = cfg["grid_shape"]
    r0 = 3 * ny * nx                              # levels 3..6: 4 M rows, ~0.27 G pairs (1 GB of indices)
    r1 = 7 * ny * nx
    ip = csr.indptr[r0:r1 + 1].to(torch.int64).cpu().numpy()
    p0, p1 = int(ip[0]), int(ip[-1])
    idx = csr.gate_indices[p0:p1].cpu().numpy()
    ip = ip - p0
    n_gates = vol.n_total_gates
    nb = search.n_binned
    orig = search.sorted_gates.view(-1, 4)[:nb, 3].contiguous().view(torch.int32).cpu().numpy()
    inv = np.zeros(n_gates, dtype=np.int32)
    inv[orig] = np.arange(nb, dtype=np.int32)
    order_b = row_mapped_order(ip)
    variants = {
        "A_orig": idx, "A_sorted": inv[idx],
        "B_orig": idx[order_b], "B_sorted": inv[idx][order_b],
    }
    packed = torch.randn(n_gates, device=dev)
    out = torch.zeros(4, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rec = {"rows": int(r1 - r0), "pairs": int(p1 - p0), "pairs_B_padded": int(len(order_b))}
    for name, arr in variants.items():
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)).to(dev)
        n = t.numel()

        def timed(fn):
            fn(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record(); b.synchronize()
                best = min(best, a.elapsed_time(b))
            return best
        g = timed(lambda: lib.gp_gather_sum(t.data_ptr(), n, packed.data_ptr(), out.data_ptr(), 16384, stream))
        s = timed(lambda: lib.gp_stream_only(t.data_ptr(), n, out.data_ptr(), 16384, stream))
        rec[name] = {"gather_ms": round(g, 4), "stream_only_ms": round(s, 4), "ns_per_64_pairs": round((g - s) * 1e6 / (n / 64), 3),
                     "gpairs_per_s": round(n / g / 1e6, 1)}
    print(json.dumps(rec, indent=1), flush=True)


if __name__ == "__main__":
    main()
