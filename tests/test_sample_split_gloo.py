"""Multi-rank sample split on CPU (gloo, world_size 2): ranks render interleaved iteration indices of the same
scene and one all-reduce sums the running sums.  The renderer here is the CPU oracle standing in for the GPU
context (the split / reduce logic is what is under test; the GPU version of the same check is
test_sample_split_equals_single_render)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, spp, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="2")
    sys.path.insert(0, HERE)
    import importlib
    import conftest
    pt = conftest.load_package()
    scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
    import oracle_binding as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.make_scene("cornell", res=(48, 48), depth=8, directory=os.path.join(out_dir, "scenes%d" % rank))
    o = orc.Oracle(pt, sc, orc.MATH_LIBM)
    for it in range(1 + rank, spp + 1, world):          # rank r renders 1+r, 1+r+R, ...
        o.trace(it)
    img = torch.from_numpy(o.image().copy())
    pb = torch.tensor([float(o.stats().pathBounces)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(img)
    dist.all_reduce(pb)
    if rank == 0:
        np.save(os.path.join(out_dir, "sum.npy"), img.numpy())
        np.save(os.path.join(out_dir, "pb.npy"), pb.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_split_equals_single_render(pt, scenes, orc, tmp_path):
    spp, world = 8, 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, spp, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sum.npy")
    sc = scenes.make_scene("cornell", res=(48, 48), depth=8)
    o = orc.Oracle(pt, sc, orc.MATH_LIBM)
    for it in range(1, spp + 1):
        o.trace(it)
    ref = o.image()
    # same samples, different fp32 summation order
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-6)
    assert float(np.load(tmp_path / "pb.npy")[0]) == float(o.stats().pathBounces)
    # the 2x2 stratification cell is iter % 4: with R in {1,2,4,8} and spp a multiple of 8 every rank covers all cells evenly
    for R in (1, 2, 4, 8):
        for r in range(R):
            cells = [(it % 4) for it in range(1 + r, 64 + 1, R)]
            assert len(cells) == 64 // R
