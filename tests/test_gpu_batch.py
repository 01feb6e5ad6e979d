"""Multi-GPU batch path on real hardware: ``batch.VolumeBatch.grid_shard`` and ``composite_max``
inside an RCCL (backend "nccl") process group, and ``bench.py``'s own rank launcher / config-5 mode.

One GPU is all a test box has, so the RCCL group has ONE rank (the collectives still go through RCCL); the
two-rank rehearsal shares the card over gloo exactly as ``bench.py --share-device`` documents.  World-size-2 logic on
CPU is covered by tests/test_batch_distributed.py.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ATOL_FRAC, REPO
from oracle import radar_grid_oracle as oracle

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_group():
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=dev)
    yield dev
    dist.destroy_process_group()


def test_volume_batch_grid_shard_under_rccl(nccl_group, tmp_path):
    """grid_shard (CSR path and fused path) for a 5-volume batch inside the RCCL group, against the oracle; the
    composite of the per-volume COLMAX planes goes through an RCCL all-reduce(MAX)."""
    import torch
    import torch.distributed as dist
    import radar_processor_amd as rg
    from radar_processor_amd import batch, synthetic
    dev = nccl_group
    assert dist.get_backend() == "nccl" and batch.rank_and_world() == (0, 1)
    shape, limits = (5, 20, 24), ((0.0, 6000.0), (-50e3, 50e3), (-60e3, 60e3))
    vols = [synthetic.make_volume(n_elev=4, n_az=90, n_gates=100, seed=20 + b, fields=("DBZH", "ZDR")) for b in range(5)]
    geom = rg.compute_grid_geometry(vols[0].gate_x, vols[0].gate_y, vols[0].gate_z, shape, limits, str(tmp_path))
    o_ip, o_idx, o_w = oracle.canonical_rows(geom.indptr, geom.gate_indices, geom.weights)
    volumes = [{k: (np.ma.getdata(v.fields[k]), np.ma.getmaskarray(v.fields[k])) for k in ("DBZH", "ZDR")} for v in vols]
    search = rg.RoiSearch(vols[0].gate_x, vols[0].gate_y, vols[0].gate_z, shape, limits, device=dev)
    for geometry in (geom, search):
        vb = batch.VolumeBatch(geometry, ["DBZH", "ZDR"], device=dev)
        events = []
        got = vb.grid_shard(volumes, events=events)
        # CSR path: 2 volumes x 2 fields per pass (3 passes); CSR-free path: 4 volumes x 2 fields (2 passes)
        assert sorted(got) == list(range(5)) and len(events) == (2 if vb.fused else 3)
        torch.cuda.synchronize()
        assert all(a.elapsed_time(b) >= 0 for a, b in events)
        for b in range(5):
            for i, name in enumerate(("DBZH", "ZDR")):
                data, mask = oracle.merge_masks(vols[b].fields[name])
                want = oracle.csr_apply(o_ip, o_idx, o_w, data, mask, shape)
                g = got[b][i].cpu().numpy()
                np.testing.assert_array_equal(np.isnan(g), np.isnan(want))
                scale = float(np.abs(data[np.isfinite(data) & ~mask]).max())
                np.testing.assert_allclose(g, want, rtol=1e-5, atol=ATOL_FRAC * scale, equal_nan=True)
        # products as a declaration (PlaneProducts) instead of a reducer: the same planes as reducing the grids by hand
        spec = rg.PlaneProducts(cappi=(3000.0,), z_min_idx=1, z_max_idx=3)
        recs = vb.grid_shard(volumes, products=spec)
        for b in range(5):
            for i in range(2):
                want_max, want_arg = rg.column_argmax(got[b][i], z_min_idx=1, z_max_idx=3)
                assert torch.equal(recs[b][i]["colmax"].view(torch.int32), want_max.view(torch.int32))
                assert torch.equal(recs[b][i]["argmax"], want_arg)
                want_cap = rg.constant_altitude_ppi(got[b][i], geom, 3000.0)
                assert torch.equal(recs[b][i]["cappi"][3000.0].view(torch.int32), want_cap.contiguous().view(torch.int32))
    planes = vb.grid_shard(volumes, products=lambda g: rg.column_max(g[0]))
    local = torch.stack([planes[b] for b in range(5)])
    mine = torch.where(torch.isnan(local), torch.full_like(local, float("-inf")), local).amax(dim=0)
    mine = torch.where(mine == float("-inf"), torch.full_like(mine, float("nan")), mine)
    comp = batch.composite_max(mine)                                       # RCCL all-reduce(MAX), one rank
    assert comp.is_cuda
    with np.errstate(all="ignore"):
        want = np.fmax.reduce(local.cpu().numpy(), axis=0)
    np.testing.assert_array_equal(comp.cpu().numpy(), want)
    t = torch.ones(8, device=dev)
    dist.all_reduce(t)                                                      # the group really is live on the device
    assert float(t.sum()) == 8.0


def _run_bench(extra, timeout=600):
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", *extra]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run(cmd, capture_output=True, timeout=timeout, env=env, cwd=REPO)
    assert res.returncode == 0, res.stderr.decode()[-4000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout.decode()
    return json.loads(lines[0])


def test_bench_config5_single_gpu_line():
    """`bench.py --config C5` (BASELINE config 5 through batch.VolumeBatch) on the config-2 grid: one JSON line with the
    contract's keys, n_gpus equal to the ranks that ran, a live kernel timing and the end-to-end leg."""
    line = _run_bench(["--gpus", "1", "--config", "C5", "--c5-grid", "C2", "--volumes-per-gpu", "3"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["config"]["volumes_total"] == 3 and line["config"]["ranks_seen_by_process_group"] == 1
    assert "config 5" in line["config"]["workload"]
    assert line["value"] > 0 and line["roofline"]["kernel_ms"] > 0 and 0 < line["roofline"]["frac"] < 1
    assert line["roofline"]["ceiling_measured"] > 2000 and "traffic_source" in line["roofline"]
    r = line["roofline"]                     # the launch-time distribution of the timed region (a mean is one draw)
    assert r["kernel_ms_min"] <= r["kernel_ms_median"] <= r["kernel_ms_max"] and r["kernel_ms_min"] <= r["kernel_ms"] <= r["kernel_ms_max"]
    assert r["launches_timed"] == r["launches_per_step"] * line["steps"] and line["config"]["record_order"] == "dispatch"
    assert line["end_to_end"]["ms_per_step"] > 0


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent starts two ranks itself (here sharing the one card over
    gloo) and the line reports the ranks the process group saw -- not a 1-GPU run labelled n_gpus=2."""
    line = _run_bench(["--gpus", "2", "--dist-backend", "gloo", "--share-device", "--config", "C5", "--c5-grid", "C2",
                       "--volumes-per-gpu", "2"])
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen_by_process_group"] == 2
    assert line["config"]["volumes_total"] == 4
    assert "skipped" in line["cpu_baseline"]            # N > 1: the host-core baseline belongs to the N = 1 line


def test_two_rank_line_carries_per_rank_kernel_times_and_the_config5_side_measurement():
    """What the driver's scaling run needs in every `--gpus N` line: each rank's median kernel time, so that an
    N-GPU efficiency can be separated from the two-cluster placement lottery of the single-field kernel, and
    north_star's batch figure (8 seeded volumes per GPU through batch.VolumeBatch, max over ranks) as `extras.c5` -- here two
    ranks sharing the one card over gloo, on the config-2 grid."""
    line = _run_bench(["--gpus", "2", "--dist-backend", "gloo", "--share-device", "--config", "C2", "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen_by_process_group"] == 2
    per_rank = line["per_rank_kernel_ms"]
    assert len(per_rank) == 2 and all(0 < v < 100 for v in per_rank)
    steps = line["per_rank_ms_per_step"]
    assert len(steps) == 2 and all(0 < v <= line["ms_per_step"] * 1.001 for v in steps)     # value is priced on the slowest rank
    c5 = line["extras"]["c5"]
    assert c5["volumes_total"] == 16 and c5["volumes_per_gpu"] == 8
    for mode in ("separate", "fused"):
        assert c5[mode]["ms_per_step"] > 0 and c5[mode]["passes_per_step"] == 2 and c5[mode]["mvoxel_s_all_gpus"] > 0
    # ... and the products mode of the config-5 line itself is recorded
    one = _run_bench(["--gpus", "1", "--config", "C5", "--c5-grid", "C2", "--volumes-per-gpu", "4", "--products", "fused"])
    assert one["config"]["products"] == "fused" and one["per_rank_kernel_ms"] == [one["roofline"]["kernel_ms_median"]]


def test_bench_refuses_mismatched_world():
    """Under a launcher (RANK / WORLD_SIZE set) --gpus must equal WORLD_SIZE: exit code 2 before any GPU work."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"], capture_output=True, env=env,
                         timeout=120, cwd=REPO)
    assert res.returncode == 2 and b"WORLD_SIZE=1 but --gpus 4" in res.stderr
