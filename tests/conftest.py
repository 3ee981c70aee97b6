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
    """Procedural stand-ins for teapot.obj / marble_bust_01.obj / old_hall_4k.hdr (small ones for tests)."""
    from hobbyraytracer_amd import api
    d = tmp_path_factory.mktemp("assets")
    api.write_teapot_obj(str(d / "teapot.obj"), 1.0)
    api.write_hall_hdr(str(d / "old_hall_4k.hdr"), 512, 256)
    api.write_bust_obj(str(d / "marble_bust_01.obj"), 0.25)   # ~6k triangles: the CPU tests stay fast
    import numpy as np
    img = np.zeros((32, 64, 3), np.uint8)
    img[:, ::8] = [255, 40, 40]
    img[:, 1::8] = [255, 40, 40]
    img[::4] = [40, 40, 255]
    api.write_image(str(d / "stripes.png"), img)               # image texture of material_zoo.yaml
    return str(d)


@pytest.fixture(scope="session")
def assets_full(built, tmp_path_factory):
    """The stand-ins at BASELINE's sizes (configs C4 / C5): teapot.obj ~6.3k triangles, marble_bust_01.obj ~100k triangles,
    old_hall_4k.hdr 4096 x 2048.  GPU tests only (the CPU suite stays with the small `assets`)."""
    from hobbyraytracer_amd import api
    d = tmp_path_factory.mktemp("assets_full")
    api.write_teapot_obj(str(d / "teapot.obj"), 1.0)
    api.write_hall_hdr(str(d / "old_hall_4k.hdr"), 4096, 2048)
    api.write_bust_obj(str(d / "marble_bust_01.obj"), 1.0)
    return str(d)


SCENES = os.path.join(ROOT, "tests", "golden", "scenes")


@pytest.fixture(scope="session")
def scenes_dir():
    return SCENES


@pytest.fixture(autouse=True)
def _no_bounds_violations(request):
    """When the suite runs on a -DHRT_DEBUG_BOUNDS build of libhrt_hip.so (tests/tools/debug_bounds.sh), every GPU test must
    end with all index checks clean.  A normal build answers "unsupported" at once."""
    yield
    if request.node.get_closest_marker("gpu") is None or not os.environ.get("HRT_HIP_LIB"):
        return
    from hobbyraytracer_amd import api
    v = api.debug_bounds_violations(0)
    assert v is None or not any(v), f"out-of-range table indices on the device (tri, prim, mat, tex, mesh, stale source, ...): {v}"
