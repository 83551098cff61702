#!/bin/bash
# Host-only robustness run of the scene / glTF / PNG loaders: random corruptions of valid inputs under
# AddressSanitizer + UBSan (no GPU, plain g++).   tools/fuzz/run.sh [rounds] [seed]
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=$ROOT/build/fuzz/work
mkdir -p "$W"
H=$ROOT/cuda-path-tracer_amd/csrc/host
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -I"$ROOT/include" "$ROOT/tools/fuzz/fuzz_host.cpp" \
    $H/scene_host.cpp $H/gltf_load.cpp $H/image_io.cpp $H/error.cpp $H/bvh_build.cpp -o "$ROOT/build/fuzz/fuzz_host"
python3 - "$W" <<'PY'
import importlib, json, pathlib, sys
import numpy as np
root = pathlib.Path(sys.argv[1]).resolve().parents[2]
sys.path.insert(0, str(root / "tests"))
import conftest
pt = conftest.load_package()
scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
import glb_util as G
import jpeg_util as J
import test_gltf_import as T
d = pathlib.Path(sys.argv[1])
T._build_test_glb(d, "seed.glb")
(d / "seed.png").write_bytes(G.write_png(np.random.default_rng(1).integers(0, 256, (24, 24, 4), dtype=np.uint8), 6, interlace=True))
(d / "seed.jpg").write_bytes(J.encode(T._smooth_image(40, 56, 2), sampling=(2, 2), restart=2)[0])
sd = scenes.scene_dict("cornell")
sd["Objects"].append(dict(TYPE="gltf", FILE="seed.glb", TRANS=[0, 1, 0], ROTAT=[0, 0, 0], SCALE=[1, 1, 1]))
(d / "seed.json").write_text(json.dumps(sd))
base = dict(Materials={"a": dict(TYPE="Diffuse", RGB=[1, 1, 1])}, Camera=scenes._camera("g", 4, 1, res=(8, 8)), Objects=[])
(d / "base.json").write_text(json.dumps(base))
PY
"$ROOT/build/fuzz/fuzz_host" "$W/seed.glb" "$W/seed.png" "$W/seed.json" "${1:-3000}" "$W" "${2:-12345}" "$W/seed.jpg"
# the same with progressive JPEG seeds (4:2:0 with restart markers, CMYK): refinement scans and end-of-band runs under corruption
"$ROOT/build/fuzz/fuzz_host" "$W/seed.glb" "$W/seed.png" "$W/seed.json" "${1:-3000}" "$W" "${2:-12345}" "$ROOT/tests/golden/progressive/rgb420_8x9_q75_restart.jpg"
"$ROOT/build/fuzz/fuzz_host" "$W/seed.glb" "$W/seed.png" "$W/seed.json" "${1:-3000}" "$W" "$((${2:-12345} + 1))" "$ROOT/tests/golden/progressive/cmyk_64x48_q75.jpg"
