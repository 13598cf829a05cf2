import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def load_package():
    """Import the hyphen-named product package pbf-sph_amd/ as module `pbf_sph_amd`."""
    if "pbf_sph_amd" in sys.modules:
        return sys.modules["pbf_sph_amd"]
    pkg_dir = os.path.join(ROOT, "pbf-sph_amd")
    spec = importlib.util.spec_from_file_location("pbf_sph_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["pbf_sph_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    m = load_package()
    m.build()
    return m


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
