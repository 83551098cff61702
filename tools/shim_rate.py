#!/usr/bin/env python3
"""pathtrace() calls per second through the file a maintainer of the reference adds (integration/pathtrace_mi355x.cpp).

    python tools/shim_rate.py [--workload dragon] [--calls 513] [--res 1920 1080] [--lookahead 128]

oracle/_ref/ref_render is the reference's own scene loader (its scene.cpp, compiled where it lies) in front of that
file and of libmi_pathtrace.so, with a headless main that calls pathtrace(nullptr, 0, iteration) once per iteration and
reads nothing else — the loop of the reference's runCuda.  The reference's loader only reads files, so the workload's
procedural stand-in model is written out as a .glb (one primitive per material, positions taken back through the
object's transform, which the loader bakes in again), its environment map as a Radiance .hdr, and the scene as the
reference's JSON.  The tool prints ref_render's own timing line: wall time per pathtrace() call with the scene's host
image up to date after every call.  Development tool (it drives test infrastructure); not part of the product.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="dragon")
    ap.add_argument("--calls", type=int, default=513)
    ap.add_argument("--res", type=int, nargs=2, default=[1920, 1080])
    ap.add_argument("--lookahead", type=int, default=128)
    ap.add_argument("--keep", help="directory to write the exported scene into (default: a temporary one)")
    a = ap.parse_args()
    import conftest
    import glb_util as G
    pt = conftest.load_package()
    scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
    render = os.path.join(ROOT, "oracle", "_ref", "ref_render")
    if not os.path.exists(render):
        sys.exit("oracle/_ref/ref_render is not built (it needs the reference tree at build time)")
    out = a.keep or tempfile.mkdtemp(prefix="shim_rate_")
    os.makedirs(out, exist_ok=True)
    cfg = scenes._ASSET_CLASS[a.workload]
    if cfg.get("textured"):
        sys.exit("only the untextured stand-in (dragon) is exported")
    sc = scenes.make_scene(a.workload)
    d = scenes.scene_dict(a.workload)
    obj = [o for o in d["Objects"] if o["TYPE"] == "gltf"][0]
    tris = sc.mesh_triangles(0)
    mats = sc.materials()
    # back through the object's transform: the loader applies it to every vertex again (src/scene.cpp:265, :531)
    M = pt.build_transforms(obj["TRANS"], obj["ROTAT"], obj["SCALE"])[0].T.astype(np.float64)
    Mi = np.linalg.inv(M)
    b = G.GlbBuilder()
    prims = []
    first_mesh_material = min(int(m) for m in np.unique(tris["materialId"]))
    for mid in sorted(int(m) for m in np.unique(tris["materialId"])):
        t = tris[tris["materialId"] == mid]
        pos = np.stack([t["v0"], t["v1"], t["v2"]], 1).reshape(-1, 3).astype(np.float64)
        nrm = np.stack([t["n0"], t["n1"], t["n2"]], 1).reshape(-1, 3).astype(np.float64)
        uv = np.stack([t["uv0"], t["uv1"], t["uv2"]], 1).reshape(-1, 2).astype(np.float32)
        lp = (np.concatenate([pos, np.ones((len(pos), 1))], 1) @ Mi.T)[:, :3].astype(np.float32)
        ln = nrm @ M[:3, :3]                       # inverse of the normal matrix (M^-1)^T is M^T
        ln = (ln / np.maximum(np.linalg.norm(ln, axis=1, keepdims=True), 1e-30)).astype(np.float32)
        m = mats[mid]
        g = b.material(pbrMetallicRoughness=dict(baseColorFactor=[float(m.color[0]), float(m.color[1]), float(m.color[2]), float(1.0 - m.transparency)],
                                                 metallicFactor=float(m.metallic), roughnessFactor=float(m.roughness)))
        assert g == mid - first_mesh_material
        prims.append(dict(attributes=dict(POSITION=b.accessor(lp, "VEC3", 5126), NORMAL=b.accessor(ln, "VEC3", 5126), TEXCOORD_0=b.accessor(uv, "VEC2", 5126)), material=g))
    b.node(root=True, mesh=b.mesh(prims))
    b.write_glb(os.path.join(out, a.workload + ".glb"))
    obj["FILE"] = a.workload + ".glb"
    if cfg.get("env"):
        pt.write_hdr(os.path.join(out, "sky.hdr"), scenes.sky_hdr(2048, 1024))
        d["EnvironmentMap"] = dict(FILE="sky.hdr", INTENSITY=cfg["env_intensity"])
    d["Camera"]["ITERATIONS"] = a.calls
    path = os.path.join(out, a.workload + "_shim.json")
    with open(path, "w") as f:
        json.dump(d, f, indent=1)
    # The reference's loadGLTFMaterial fills a `Material mat;` that is never zeroed (src/scene.cpp:96): index of refraction,
    # the specular block and every subsurface field of a glTF material are whatever the stack held, so two processes need
    # not load the same scene (ref_render prints a hash of what the loader produced).  The two modes are compared bit for
    # bit where the scene is under control (tests/test_gpu_parity.py::test_lookahead_calls_bit_exact, tools/diag/lookahead_full.py);
    # here the sums are compared only when the material bytes of the two runs agree.
    runs = []
    for mode in ("strict", "queued", "queued"):
        env = dict(os.environ)
        env.pop("LD_BIND_NOW", None)
        env["PT_LOOKAHEAD"] = str(a.lookahead)
        env["PT_HOST_COPY"] = mode
        env["REF_RENDER_PIN_MATERIALS"] = "1"
        if os.environ.get("SHIM_STREAMS"):
            env["PT_LOOKAHEAD_STREAMS"] = os.environ["SHIM_STREAMS"]
        r = subprocess.run([render, path, str(a.calls), os.path.join(out, "sum.bin"), str(a.res[0]), str(a.res[1])],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
        err = r.stderr.decode(errors="replace")
        if r.returncode != 0:
            sys.exit("ref_render failed (%d): %s" % (r.returncode, err[-800:]))
        line = [l for l in err.splitlines() if l.startswith("ref_render:") and "hashes" not in l][-1]
        hashes = [l for l in err.splitlines() if "hashes" in l][-1]
        mat_hash = hashes.split("materials ")[1].split(",")[0]
        raw = open(os.path.join(out, "sum.bin"), "rb").read()
        img = np.frombuffer(raw[8:], np.float32).copy()
        runs.append((mode, mat_hash, img))
        print("%s | %d triangles through the reference's loader, look-ahead %d, material bytes %s, mean radiance of the sum / calls = %.5f, finite: %s"
              % (line, len(tris), a.lookahead, mat_hash, float(img.mean()) / a.calls, bool(np.isfinite(img).all())), flush=True)
    for i in range(len(runs)):
        for j in range(i + 1, len(runs)):
            if runs[i][1] == runs[j][1]:
                same = bool((runs[i][2].view(np.uint32) == runs[j][2].view(np.uint32)).all())
                print("runs %d (%s) and %d (%s) loaded the same material bytes: sums bit-identical: %s" % (i, runs[i][0], j, runs[j][0], same))
                if not same:
                    sys.exit("two runs of one scene differ")


if __name__ == "__main__":
    main()
