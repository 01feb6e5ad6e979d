#!/usr/bin/env python3
"""Experiment: does renumbering the gates in cell-sorted order (so that the pairs of a row gather from a contiguous
run of the packed field) speed up rg_csr_apply_f32?  Same CSR bytes, same pair order, bit-identical results."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import radar_processor_amd as rg
    from radar_processor_amd import _native, synthetic
    from radar_processor_amd.gridding import CsrGridder
    name = sys.argv[1] if len(sys.argv) > 1 else "METRIC"
    cfg = synthetic.CONFIGS[name]
    vol = synthetic.make_volume(cfg["n_elev"], cfg["n_az"], cfg["n_gates"], seed=0, fields=("DBZH",))
    search = rg.RoiSearch(vol.gate_x, vol.gate_y, vol.gate_z, cfg["grid_shape"], cfg["grid_limits"])
    csr = search.build_csr("barnes2")
    geom = rg.GridGeometry.from_device(cfg["grid_shape"], cfg["grid_limits"], csr, 17000.0)
    dev = search.dev
    field = vol.fields["DBZH"]
    f_t = torch.from_numpy(np.ascontiguousarray(np.ma.getdata(field))).to(dev)
    m_t = torch.from_numpy(np.ma.getmaskarray(field).astype(np.uint8)).to(dev)
    n_gates = f_t.numel()
    g = CsrGridder(geom, n_gates, 1, device=dev)
    g.pack([f_t], [m_t])
    n_vox = g.n_vox
    out_a = torch.empty(n_vox, dtype=torch.float32, device=dev)
    out_b = torch.empty(n_vox, dtype=torch.float32, device=dev)
    lib = _native.load_library()

    nb = search.n_binned
    orig = search.sorted_gates.view(-1, 4)[:nb, 3].contiguous().view(torch.int32).long()
    inv = torch.zeros(n_gates, dtype=torch.int32, device=dev)
    inv[orig] = torch.arange(nb, dtype=torch.int32, device=dev)
    new_idx = torch.empty_like(csr.gate_indices)
    step = 1 << 28
    for a in range(0, csr.n_pairs, step):
        new_idx[a:a + step] = inv[csr.gate_indices[a:a + step].long()]
    packed_perm = torch.zeros_like(g.packed)
    packed_perm[:nb] = g.packed[orig]

    def run(idx, packed, out):
        _native.check(lib.rg_csr_apply_f32(_native.ptr(csr.indptr), int(csr.is_i64), _native.ptr(idx),
                                           _native.ptr(csr.weights), n_vox, csr.n_pairs, _native.ptr(packed), 1, 1,
                                           n_gates, float("nan"), _native.ptr(out), _native.stream_ptr()), "apply")

    def timeit(idx, packed, out, reps=10):
        run(idx, packed, out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            run(idx, packed, out)
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / reps

    rec = {"pairs": csr.n_pairs}
    for rnd in range(2):
        rec[f"orig_ms_{rnd}"] = round(timeit(csr.gate_indices, g.packed, out_a), 3)
        rec[f"renumbered_ms_{rnd}"] = round(timeit(new_idx, packed_perm, out_b), 3)
    rec["bit_identical"] = bool(torch.equal(out_a.view(torch.int32), out_b.view(torch.int32)))
    print(name, json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
