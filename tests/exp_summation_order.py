#!/usr/bin/env python3
"""CPU experiment behind the row-wise kernel's order of summation (DESIGN.md, K1c "Why two chains"); not a test.

    python tests/exp_summation_order.py [reference|exact]

Replays the kernel's lane split on the REFERENCE's CSR fixtures (g3, config-2 geometry) in NumPy with different ways of
adding a lane's float32 products -- one sequential chain (round 2), C chains by batch slot, a record-local sum first,
float64 everywhere -- and prints, per field, the worst and the RMS relative error on the voxels with |value| > 1e-3 *
max|field|: against the reference's gridded fixtures (which carry the reference's own float32 rounding) or against the
exact float64 sums of the same float32 products.  Lives under tests/ because it imports the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import golden_names, grid_spec, load_golden, reference_indices, volume_for   # noqa: E402
from oracle import radar_grid_oracle as oracle                                              # noqa: E402

MODES = ("seq", "chains2", "chains3", "recsum1", "recsum2", "pair64")


def lane_sums(m, mode, kpre=3):
    """m: float32 [lanes, trips, 3] (a lane's records in order, three pair slots each, zeros where there is no pair).
    Returns the lane totals in float64 (already folded per lane; the caller folds the lanes)."""
    lanes, ntrip, _ = m.shape
    f32 = np.float32

    def seq(x):                                    # strictly sequential float32 sum along the last axis
        return np.add.accumulate(x.reshape(lanes, -1), axis=1, dtype=f32)[:, -1] if x.size else np.zeros(lanes, f32)

    if mode == "seq":
        return seq(m)
    if mode.startswith("chains"):
        c_n = int(mode[6:])
        tot = None
        for c in range(c_n):
            idx = [t for t in range(ntrip) if (t % kpre) % c_n == c]
            part = seq(m[:, idx]) if idx else np.zeros(lanes, f32)
            tot = part if tot is None else (tot + part).astype(f32)
        return tot
    if mode.startswith("recsum"):
        c_n = int(mode[6:])
        rsum = ((m[:, :, 0] + m[:, :, 1]).astype(f32) + m[:, :, 2]).astype(f32)
        tot = None
        for c in range(c_n):
            idx = [t for t in range(ntrip) if (t % kpre) % c_n == c]
            part = np.add.accumulate(rsum[:, idx], axis=1, dtype=f32)[:, -1] if idx else np.zeros(lanes, f32)
            tot = part if tot is None else (tot + part).astype(f32)
        return tot
    return m.astype(np.float64).reshape(lanes, -1).sum(axis=1)        # pair64: exact


def rowwise(indptr, gidx, w_all, vals, excl, shape, mode, target=4):
    nz, ny, nx = shape
    out = np.full(nz * ny * nx, np.nan, dtype=np.float64)
    nsx = (nx + 63) // 64
    sb, se = nx // nsx, nx % nsx
    zero = np.float32(0)
    for line in range(nz * ny):
        x0 = 0
        for sx in range(nsx):
            nrows = sb + (1 if sx < se else 0)
            r0 = line * nx + x0
            x0 += nrows
            seg_b = int(indptr[r0])
            span = int(indptr[r0 + nrows]) - seg_b
            if span == 0:
                continue
            need = (span // (3 * nrows) + 1 + target - 1) // target
            lanes = 1 << min(0 if need <= 1 else (need - 1).bit_length(), 6)
            for r in range(r0, r0 + nrows):
                ps, pe = int(indptr[r]), int(indptr[r + 1])
                if pe == ps:
                    continue
                rs = ps - seg_b
                o = np.arange(rs, pe - seg_b)
                q = o // 3 - rs // 3
                lane, trip, j = q % lanes, q // lanes, o % 3
                g, w = gidx[ps:pe], w_all[ps:pe]
                good = ~excl[g]
                prod = np.where(good, w * vals[g], zero).astype(np.float32)
                wgt = np.where(good, w, zero).astype(np.float32)
                ntrip = int(trip.max()) + 1
                mp = np.zeros((lanes, ntrip, 3), np.float32)
                mw = np.zeros((lanes, ntrip, 3), np.float32)
                mp[lane, trip, j] = prod
                mw[lane, trip, j] = wgt
                tots = []
                for m in (mp, mw):
                    t = lane_sums(m, mode)
                    if mode == "pair64":
                        tots.append(float(t.sum()))
                        continue
                    k = 1
                    while k < lanes:                      # float32 xor butterfly
                        t = (t + t[np.arange(lanes) ^ k]).astype(np.float32)
                        k <<= 1
                    tots.append(float(t[0]))
                if tots[1] > 0:
                    out[r] = np.float32(tots[0] / tots[1])
    return out.reshape(shape)


def main():
    against = sys.argv[1] if len(sys.argv) > 1 else "reference"
    res = {}
    for name in golden_names("g3_c2_r150_barnes2") + golden_names("g3_c2_r060") + golden_names("g3_c2_r235"):
        meta, ref = load_golden(name)
        vol = volume_for(meta)
        shape, _ = grid_spec(meta)
        idx = reference_indices(name, meta, ref)
        ip = np.asarray(ref["indptr"], np.int64)
        wts = np.asarray(ref["weights"], np.float32)
        for fname in ("ZDR", "DBZH"):
            if fname not in meta["fields"]:
                continue
            data, mask = oracle.merge_masks(vol.fields[fname])
            data32 = np.asarray(data, np.float32)
            scale = float(np.nanmax(np.abs(data[~mask & np.isfinite(data)])))
            want = ref[f"grid_{fname}"].astype(np.float64)
            filled = np.isfinite(want)
            if against == "exact":
                want = rowwise(ip, idx, wts, data32, mask, shape, "pair64")
            mag = np.abs(want[filled])
            sig = mag > 1e-3 * scale
            for mode in MODES:
                got = rowwise(ip, idx, wts, data32, mask, shape, mode)
                rel = (np.abs(got[filled] - want[filled]) / mag)[sig]
                prev = res.get((fname, mode), (0.0, 0.0))
                res[(fname, mode)] = (max(prev[0], float(rel.max())), max(prev[1], float(np.sqrt((rel ** 2).mean()))))
        print(name, against, {f"{k[0]}/{k[1]}": f"max {v[0]:.2e} rms {v[1]:.2e}" for k, v in res.items()}, flush=True)


if __name__ == "__main__":
    main()
