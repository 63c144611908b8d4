import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """Import the product package from its hyphenated directory as `olap_in_memory_amd`."""
    name = "olap_in_memory_amd"
    if name in sys.modules:
        return sys.modules[name]
    # torch bundles its own HIP runtime: a process that uses both must load torch's FIRST, so that
    # libolapgpu.so binds to the runtime already in the process (two runtimes cannot share the GPU)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    pkg_dir = os.path.join(ROOT, "olap-in-memory_amd")
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_package()
