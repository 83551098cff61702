"""pytest configuration: markers, package loading, oracle access.

GPU tests are marked `gpu` and call the product only through the C ABI
(libmi_pathtrace.so).  The CPU oracle (oracle/liboracle.so) is loaded here, and only
here, as the checker.
"""
import importlib
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_package():
    name = "cuda_path_tracer_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "cuda-path-tracer_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _ensure_built():
    lib = os.path.join(ROOT, "cuda-path-tracer_amd", "csrc", "libmi_pathtrace.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.dirname(lib), "-j8"])
    if not os.path.exists(orc):
        subprocess.check_call(["make", "-C", os.path.dirname(orc)])


@pytest.fixture(scope="session")
def pt():
    _ensure_built()
    return load_package()


@pytest.fixture(scope="session")
def scenes(pt):
    return importlib.import_module("cuda_path_tracer_amd.scenes")


@pytest.fixture(scope="session")
def orc(pt):
    import oracle_binding
    return oracle_binding


@pytest.fixture(scope="session")
def gpu_ok():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    return True
