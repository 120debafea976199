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
    """import the product package (directory name has a hyphen) as `thaler_study_amd`"""
    if "thaler_study_amd" in sys.modules:
        return sys.modules["thaler_study_amd"]
    path = os.path.join(ROOT, "thaler-study_amd")
    spec = importlib.util.spec_from_file_location(
        "thaler_study_amd", os.path.join(path, "__init__.py"), submodule_search_locations=[path])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["thaler_study_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_package()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """without a GPU the -m gpu tests are skipped, not errors (a plain `pytest` stays green on CPU)"""
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
