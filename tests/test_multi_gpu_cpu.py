"""The N > 1 path on the CPU: two processes over gloo render interleaved row stripes (with the CPU build of the device
functions standing in for the GPU) and gather them to rank 0 through gi_raytracer_amd.sharding.FrameGather -- the same code
bench.py runs over RCCL.  The gathered frame must equal the single-process frame bit for bit."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import emul_lib as el
    import parity_checks as pc
    from gi_raytracer_amd.sharding import FrameGather

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, spp, stripe_h = 40, 27, 2, 4
    scene = pc.load_scene("caustics")
    rt = el.EmulRayTracer().setScene(scene)
    rt.tracePhotons(1500)                       # every rank emits the same photons (keyed RNG): identical maps, no exchange
    fg = FrameGather(torch, dist, w, h, stripe_h, rank, world, "cpu", torch.float64)
    part = rt.run(w, h, min_samples=spp, max_samples=spp, stripe_h=stripe_h, rank=rank, world=world)
    assert len(part) == len(fg.rows[rank])
    fg.local[: len(part)] = torch.from_numpy(part)
    dist.barrier()
    frame = fg.gather()
    if rank == 0:
        full = rt.run(w, h, min_samples=spp, max_samples=spp)
        np.save(out_path, np.stack([frame.numpy(), full]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_stripes_gathered_over_gloo_equal_single_process_frame(tmp_path, world):
    import torch.multiprocessing as mp
    out = str(tmp_path / "frames.npy")
    port = 29650 + world
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    frames = np.load(out)
    assert np.array_equal(frames[0], frames[1])


def test_stripe_partition_covers_every_row_once():
    from gi_raytracer_amd.sharding import stripe_rows, max_local_rows
    for h, sh, world in [(1080, 16, 8), (1080, 16, 3), (27, 4, 2), (5, 16, 8), (2160, 16, 8)]:
        rows = np.concatenate([stripe_rows(h, sh, r, world) for r in range(world)])
        assert sorted(rows.tolist()) == list(range(h))
        assert max_local_rows(h, sh, world) >= h // world
