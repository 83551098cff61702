"""pt_intersect_rays_noalloc (staging in the library's static device array) against pt_intersect_rays:
same kernels, so every field of every hit record must be identical, bit for bit — on two calls in a
row, the second with fewer rays (the array keeps the first call's contents), and across the limit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rays(seed, n, center=(0.0, 4.0, 0.0), radius=12.0):
    rng = np.random.RandomState(seed)
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    o = np.asarray(center) + radius * v
    d = np.asarray(center) + rng.uniform(-4, 4, size=(n, 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o.astype(np.float32), d.astype(np.float32)


def _identical(a, b):
    return a.tobytes() == b.tobytes()


def test_no_alloc_entry_point_matches_allocating_one(pt, scenes, gpu_ok):
    # 96x96 path state: both calls below need several chunks through it
    sc = scenes.make_scene("dragon", res=(96, 96), grid=(64, 65), env_size=(64, 32), tex_size=16)
    tr = pt.Tracer(sc)
    ro, rd = _rays(21, 60000)
    import torch
    calls0 = pt.lib.pt_resource_calls()
    want = tr.intersect(ro, rd)
    calls1 = pt.lib.pt_resource_calls()
    assert calls1 > calls0, "the allocating entry point is expected to allocate its staging buffers"
    tr.intersect_noalloc(ro, rd)                     # first use: nothing may be left to set up lazily either
    free0 = torch.cuda.mem_get_info()[0]
    calls1 = pt.lib.pt_resource_calls()
    got = tr.intersect_noalloc(ro, rd)
    # the property the entry point exists for: no device memory, pinned memory, event or stream was created or
    # released by the call (the library counts every such call it makes), and the device's free memory did not move
    assert pt.lib.pt_resource_calls() == calls1
    assert torch.cuda.mem_get_info()[0] == free0
    assert (want["triangle"] >= 0).sum() > 5000 and (want["t"] < 0).sum() > 100
    assert _identical(got, want)
    # second call, smaller input: entries beyond it must stay untouched, the first ones must be this call's
    ro2, rd2 = _rays(22, 7001)
    out = np.zeros(9000, dtype=pt.HIT_DTYPE)
    out["geom"] = 12345
    tr.intersect_noalloc(ro2, rd2, out=out[:7001])
    assert _identical(out[:7001], tr.intersect(ro2, rd2))
    assert np.all(out["geom"][7001:] == 12345)
    # primitives only (no traversal kernel in the chain), a second context on the same GPU
    sc2 = scenes.make_scene("cornell", res=(64, 64))
    tr2 = pt.Tracer(sc2)
    ro3, rd3 = _rays(23, 5000, center=(0.0, 5.0, 0.0), radius=9.0)
    assert _identical(tr2.intersect_noalloc(ro3, rd3), tr2.intersect(ro3, rd3))
    # the first context still answers the same
    assert _identical(tr.intersect_noalloc(ro, rd), want)
    # limits: the maximum is accepted, one more is refused, nothing is an empty success
    n = pt.PT_NOALLOC_MAX_RAYS
    rb, db = _rays(24, n + 1)
    full = tr.intersect_noalloc(rb[:n], db[:n])
    assert _identical(full, tr.intersect(rb[:n], db[:n]))
    calls2 = pt.lib.pt_resource_calls()
    with pytest.raises(pt.PtError, match="exceed"):
        tr.intersect_noalloc(rb, db)
    assert len(tr.intersect_noalloc(rb[:0], db[:0])) == 0
    tr.set_option(pt.PT_OPT_COUNTERS, 1)            # the counted kernels and stage timing do not make it allocate either
    tr.set_option(pt.PT_OPT_STAGE_TIMING, 1)
    assert _identical(tr.intersect_noalloc(ro2, rd2), out[:7001])
    assert pt.lib.pt_resource_calls() == calls2
