import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name)))
            if name == "jpeg_files.npz":      # the 4:2:2 file fixture of round 2 lives in its own archive
                cache[name].update(dict(np.load(os.path.join(GOLDEN, "jpeg_file_422.npz"))))
                cache[name].update(dict(np.load(os.path.join(GOLDEN, "jpeg_file_411.npz"))))   # round 3: h4v1 (4:1:1, with DRI) and h1v4 files
            if name == "hevc_file.npz":       # tag "e": the same kind of picture, decoded by the reference from a .heic container
                cache[name].update(dict(np.load(os.path.join(GOLDEN, "heic_file.npz"))))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def ffo():
    import oracle_lib
    return oracle_lib.ffo()


@pytest.fixture(autouse=True)
def _ffhip_switches_reread():
    """The library reads its FFHIP_* switches once per process (ffhip_reload_env makes it read them again): a test that flips
    one with monkeypatch.setenv calls capi.reload_env() itself; this puts the library back after monkeypatch has undone it."""
    yield
    from ffpic_amd import capi
    capi.reload_env()
