#!/usr/bin/env python3
"""Distribution of distinct gates per 256-row chunk (sizing of the compact CSR's LDS window)."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    name = sys.argv[1] if len(sys.argv) > 1 else "METRIC"
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    cfg = synthetic.CONFIGS[name]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"])
    csr = search.build_csr("barnes2")
    dev = search.dev
    n_vox = csr.n_vox
    n_chunks = (n_vox + rows - 1) // rows
    counts = torch.zeros(n_chunks, dtype=torch.int64, device=dev)
    pairs = torch.zeros(n_chunks, dtype=torch.int64, device=dev)
    slab = 8192
    for c0 in range(0, n_chunks, slab):
        c1 = min(n_chunks, c0 + slab)
        r0, r1 = c0 * rows, min(n_vox, c1 * rows)
        ip = csr.indptr[r0:r1 + 1].to(torch.int64)
        p0, p1 = int(ip[0]), int(ip[-1])
        if p1 == p0:
            continue
        chunk_of_row = torch.arange(r1 - r0, device=dev, dtype=torch.int64) // rows
        chunk_of_pair = torch.repeat_interleave(chunk_of_row, ip[1:] - ip[:-1], output_size=p1 - p0)
        key = (chunk_of_pair << 32) | csr.gate_indices[p0:p1].to(torch.int64)
        uniq = torch.unique(key)
        counts[c0:c1] = torch.bincount(uniq >> 32, minlength=c1 - c0)
        pairs[c0:c1] = torch.bincount(chunk_of_pair, minlength=c1 - c0)
    c = counts.cpu().numpy()
    p = pairs.cpu().numpy()
    rec = {"rows_per_chunk": rows, "chunks": int(n_chunks), "dict_total": int(c.sum()), "pairs": int(p.sum()),
           "dict_bytes_per_pair": round(4 * c.sum() / p.sum(), 3),
           "max": int(c.max()), "p50": int(np.percentile(c, 50)), "p90": int(np.percentile(c, 90)),
           "p99": int(np.percentile(c, 99)), "p999": int(np.percentile(c, 99.9))}
    for lim in (2048, 4096, 8192, 16384):
        over = c > lim
        rec[f"chunks_over_{lim}"] = int(over.sum())
        rec[f"pairs_frac_over_{lim}"] = round(float(p[over].sum() / p.sum()), 5)
    print(name, json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
