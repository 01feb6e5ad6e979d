#!/usr/bin/env python3
"""How compressible are the 16-bit dictionary positions of the compact copy?  Per row, consecutive pairs mostly refer to
consecutive gates of a ray, and the dictionary numbers gates in order of first mention -- so position deltas inside a row
should be small.  Prints, for one configuration, the histogram of in-row deltas and the share of row-aligned groups of 4 pairs
whose three deltas all lie in {1}, in [0, 3] and in [-4, 3] (what a 16-byte record of FOUR pairs -- 4 x 26-bit weights + a
16-bit base + three 2- or 3-bit deltas -- could hold instead of three).  Measurement only (round 4)."""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import synthetic
    cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    cfg = synthetic.CONFIGS[cfg_name]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    with tempfile.TemporaryDirectory() as tmp:
        geom = rg.compute_grid_geometry(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"], tmp, layout="csr")
    dev = torch.device("cuda", 0)
    csr = geom.device_csr(dev)
    compact = geom.device_compact(dev)
    n_vox = csr.n_vox
    out = {"config": cfg_name, "pairs": csr.n_pairs}
    hist = torch.zeros(33, dtype=torch.int64, device=dev)          # deltas -16 .. +16 (clamped)
    g4 = dict(groups=0, all_plus1=0, all_0_3=0, all_m4_3=0, all_m8_7=0)
    rows_per_slab = 2_000_000
    for r0 in range(0, n_vox, rows_per_slab):
        r1 = min(n_vox, r0 + rows_per_slab)
        ip = csr.indptr[r0:r1 + 1].to(torch.int64)
        p0, p1 = int(ip[0]), int(ip[-1])
        if p1 - p0 < 2:
            continue
        pos = compact.local_idx[p0:p1].to(torch.int64) & 0xFFFF
        lens = ip[1:] - ip[:-1]
        row = torch.repeat_interleave(torch.arange(r1 - r0, device=dev), lens, output_size=p1 - p0)
        k = torch.arange(p1 - p0, device=dev) - (ip[:-1] - p0)[row]                 # pair's index inside its row
        d = pos[1:] - pos[:-1]
        same = row[1:] == row[:-1]
        hist += torch.bincount((d[same].clamp(-16, 16) + 16), minlength=33)
        # row-aligned groups of four: pairs k = 4j .. 4j+3 of one row
        start = (k % 4 == 0) & (k + 3 < lens[row])
        idx = torch.nonzero(start).squeeze(1)
        if idx.numel():
            dd = torch.stack([pos[idx + 1] - pos[idx], pos[idx + 2] - pos[idx + 1], pos[idx + 3] - pos[idx + 2]], 1)
            g4["groups"] += int(idx.numel())
            g4["all_plus1"] += int((dd == 1).all(1).sum())
            g4["all_0_3"] += int(((dd >= 0) & (dd <= 3)).all(1).sum())
            g4["all_m4_3"] += int(((dd >= -4) & (dd <= 3)).all(1).sum())
            g4["all_m8_7"] += int(((dd >= -8) & (dd <= 7)).all(1).sum())
    h = hist.cpu().numpy()
    out["delta_hist"] = {str(i - 16): int(v) for i, v in enumerate(h) if v}
    out["share_delta_plus1"] = float(h[17] / max(h.sum(), 1))
    out["groups_of_4"] = {k: (v if k == "groups" else round(v / max(g4["groups"], 1), 4)) for k, v in g4.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
