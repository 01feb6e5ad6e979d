"""N>1 path on CPU: world_size-2 gloo process groups (real rendezvous over 127.0.0.1) exercising the sharding
rule, the result collection and the NaN-aware composite all-reduce that bench.py / batch.py use on RCCL.
The per-volume work inside the workers is the CPU oracle (a stand-in for the GPU gridder: tests may use it)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import REPO
from radar_processor_amd import batch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    import torch
    sys.path.insert(0, os.environ["RG_REPO"])
    from radar_processor_amd import batch
    from oracle import radar_grid_oracle as oracle

    assert batch.init_distributed("gloo")
    rank, world = batch.rank_and_world()
    n_vol = int(os.environ["RG_NVOL"])
    shape = (3, 6, 5)
    n_vox = int(np.prod(shape))
    rng = np.random.default_rng(1234)               # same geometry on every rank (replicated)
    lengths = rng.integers(0, 6, size=n_vox)
    indptr = np.zeros(n_vox + 1, dtype=np.int64); np.cumsum(lengths, out=indptr[1:])
    idx = rng.integers(0, 50, size=int(indptr[-1])).astype(np.int32)
    w = (rng.random(idx.shape[0]) + 0.1).astype(np.float32)

    def work(b):                                      # volume b: seeded by b, independent of the rank
        r = np.random.default_rng(b)
        data = r.normal(10, 5, size=50).astype(np.float32)
        mask = r.random(50) < 0.2
        grid = oracle.csr_apply(indptr, idx, w, data, mask, shape)
        return oracle.column_max(grid, 0, shape[0] - 1)

    local = batch.run_sharded(n_vol, work)
    assert sorted(local) == batch.shard_indices(n_vol, rank, world)
    planes = batch.gather_results(local, n_vol, dst=0)
    mine = [local[b] for b in sorted(local)]
    stack = np.stack(mine) if mine else np.full((1, shape[1], shape[2]), np.nan, dtype=np.float32)
    with np.errstate(all="ignore"):
        own = np.fmax.reduce(stack, axis=0).astype(np.float32)
    comp = batch.composite_max(torch.from_numpy(own)).numpy()
    if rank == 0:
        np.savez(os.environ["RG_OUT"], planes=np.stack(planes), composite=comp)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
""")


def _run_world(tmp_path, world, n_vol):
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / f"out_{world}_{n_vol}.npz"
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RG_REPO=REPO, RG_NVOL=str(n_vol), RG_OUT=str(out), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    with np.load(out) as z:
        return z["planes"], z["composite"]


def test_shard_indices_cover_every_item_once():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            seen = sorted(b for r in range(world) for b in batch.shard_indices(n, r, world))
            assert seen == list(range(n))
            sizes = [len(batch.shard_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert batch.shard_indices(64, 3, 8) == list(range(3, 64, 8))     # volume b -> GPU b mod 8
    with pytest.raises(ValueError):
        batch.shard_indices(4, 2, 2)


def test_single_process_paths():
    import torch
    plane = torch.tensor([[1.0, float("nan")], [float("nan"), -2.0]])
    out = batch.composite_max(plane)
    assert torch.equal(torch.isnan(out), torch.isnan(plane)) and out[0, 0] == 1.0 and out[1, 1] == -2.0
    assert batch.rank_and_world() == (0, 1)
    assert batch.gather_results({0: "a", 1: "b"}, 2) == ["a", "b"]
    assert batch.run_sharded(3, lambda b: b * b) == {0: 0, 1: 1, 2: 4}


@pytest.mark.parametrize("n_vol", [5, 8])
def test_world2_matches_single_process(tmp_path, n_vol):
    """Two gloo ranks produce exactly the planes a single process computes, in volume order, and the composite
    equals np.fmax over all volumes."""
    planes2, comp2 = _run_world(tmp_path, 2, n_vol)
    # single-process truth computed here with the same recipe
    from oracle import radar_grid_oracle as oracle
    shape = (3, 6, 5)
    n_vox = int(np.prod(shape))
    rng = np.random.default_rng(1234)
    lengths = rng.integers(0, 6, size=n_vox)
    indptr = np.zeros(n_vox + 1, dtype=np.int64)
    np.cumsum(lengths, out=indptr[1:])
    idx = rng.integers(0, 50, size=int(indptr[-1])).astype(np.int32)
    w = (rng.random(idx.shape[0]) + 0.1).astype(np.float32)
    truth = []
    for b in range(n_vol):
        r = np.random.default_rng(b)
        data = r.normal(10, 5, size=50).astype(np.float32)
        mask = r.random(50) < 0.2
        truth.append(oracle.column_max(oracle.csr_apply(indptr, idx, w, data, mask, shape), 0, shape[0] - 1))
    truth = np.stack(truth)
    np.testing.assert_array_equal(planes2, truth)
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(comp2, np.fmax.reduce(truth, axis=0))
