"""Multi-rank sample split on CPU (gloo, world_size 2 and 4).

What is under test is everything bench.py does around the renderer when several ranks run: every rank prepares the
scene at the same moment FROM ONE SHARED DIRECTORY (a rank must never read a scene file another rank is still
writing), bench.iteration_range / bench.per_rank_iterations decide which iteration indices a rank renders (weak and
strong split), and one all-reduce sums the running sums.  The renderer here is the CPU oracle standing in for the GPU
context; the GPU version of the same check is test_gpu_parity.py::test_sample_split_equals_single_render."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, steps, per_rank, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="2")
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import importlib
    import conftest
    pt = conftest.load_package()
    scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
    import bench
    import oracle_binding as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()                                      # all ranks reach the scene set-up together
    shared = os.path.join(out_dir, "scenes")            # ONE directory for every rank
    for _ in range(20):                                 # many rounds of write + load: a torn file would fail the parse
        sc = scenes.make_scene("cornell", res=(48, 48), depth=8, directory=shared)
    duck = scenes.make_duck_scene(res=(8, 8), depth=2, directory=shared, env_size=(16, 8))
    assert int(duck.desc.contents.meshes[0].triangleCount) == 4212
    o = orc.Oracle(pt, sc, orc.MATH_LIBM)
    begin, end, stride = bench.iteration_range(rank, world, per_rank, 0, steps)
    mine = list(range(begin, end, stride))
    for it in mine:
        o.trace(it)
    img = torch.from_numpy(o.image().copy())
    pb = torch.tensor([float(o.stats().pathBounces)], dtype=torch.float64)
    seen = torch.zeros(steps * per_rank * world + 2, dtype=torch.int32)
    seen[mine] += 1
    dist.barrier()
    dist.all_reduce(img)
    dist.all_reduce(pb)
    dist.all_reduce(seen)
    if rank == 0:
        np.save(os.path.join(out_dir, "sum.npy"), img.numpy())
        np.save(os.path.join(out_dir, "pb.npy"), pb.numpy())
        np.save(os.path.join(out_dir, "seen.npy"), seen.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,scaling", [(2, "weak"), (4, "strong")])
def test_sample_split_equals_single_render(pt, scenes, orc, tmp_path, world, scaling):
    sys.path.insert(0, ROOT)
    import bench
    total = 8
    per_rank = bench.per_rank_iterations(scaling, world, 4, total)       # weak: 4 per rank and step; strong: 8 / 4 = 2
    steps = 1
    spp = per_rank * world * steps
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, steps, per_rank, str(tmp_path)), nprocs=world, join=True)
    seen = np.load(tmp_path / "seen.npy")
    assert seen[0] == 0 and np.all(seen[1:spp + 1] == 1) and np.all(seen[spp + 1:] == 0)      # 1 .. spp, each exactly once
    got = np.load(tmp_path / "sum.npy")
    sc = scenes.make_scene("cornell", res=(48, 48), depth=8)
    o = orc.Oracle(pt, sc, orc.MATH_LIBM)
    for it in range(1, spp + 1):
        o.trace(it)
    ref = o.image()
    # same samples, different fp32 summation order
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-6)
    assert float(np.load(tmp_path / "pb.npy")[0]) == float(o.stats().pathBounces)


def test_split_rules():
    sys.path.insert(0, ROOT)
    import bench
    # the union over ranks of a step's iteration indices is a contiguous block, whatever the rank count; consecutive
    # steps continue where the last one ended
    for world in (1, 2, 4, 8):
        for per_rank in (1, 3, 64):
            got = []
            for step in range(3):
                for r in range(world):
                    b, e, s = bench.iteration_range(r, world, per_rank, step, step + 1)
                    block = list(range(b, e, s))
                    assert len(block) == per_rank
                    got += block
            assert sorted(got) == list(range(1, 3 * per_rank * world + 1))
    # strong split: total samples per pixel fixed, a multiple of lcm(4, ranks), so that the four jitter cells
    # (iteration % 4) of the union stay balanced and every rank gets the same share
    assert bench.per_rank_iterations("strong", 8, 0, 64) == 8
    assert bench.per_rank_iterations("strong", 2, 0, 512) == 256
    assert bench.per_rank_iterations("weak", 8, 128, 64) == 128
    for world, bad in ((8, 60), (2, 6), (1, 2), (4, 0)):
        with pytest.raises(SystemExit):
            bench.per_rank_iterations("strong", world, 0, bad)
    for world in (1, 2, 4, 8):
        cells = [it % 4 for r in range(world) for it in range(*bench.iteration_range(r, world, 64 // world, 0, 1))]
        assert [cells.count(c) for c in range(4)] == [16, 16, 16, 16]
    # every rank prepares its scene file in a directory of its own
    assert len({bench.scene_directory(r) for r in range(8)}) == 8
