import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Builds every native artefact once per session (no-op when up to date)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def assets(built, tmp_path_factory):
    """Procedural stand-ins for teapot.obj / old_hall_4k.hdr (small env map for tests)."""
    from hobbyraytracer_amd import api
    d = tmp_path_factory.mktemp("assets")
    api.write_teapot_obj(str(d / "teapot.obj"), 1.0)
    api.write_hall_hdr(str(d / "old_hall_4k.hdr"), 512, 256)
    return str(d)


SCENES = os.path.join(ROOT, "tests", "golden", "scenes")


@pytest.fixture(scope="session")
def scenes_dir():
    return SCENES
