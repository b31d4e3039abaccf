import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "webgpu-pathtracer_amd", "py"), os.path.join(ROOT, "oracle"), ROOT,
          os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure libmi3pt.so and the oracle exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(scope="session")
def orc(built):
    import pt_oracle
    pt_oracle.lib()
    return pt_oracle


@pytest.fixture(scope="session")
def demo(built):
    from mi3pt_host import scenes
    sc = scenes.demo_scene()
    sc.build_bvh()
    return sc


@pytest.fixture(scope="session")
def env():
    from mi3pt_host import scenes
    return scenes.synthetic_env()


@pytest.fixture(scope="session")
def gpu_ctx(built):
    """One context for the whole GPU session (tests run in one process)."""
    from mi3pt_host import capi
    ctx = capi.Context(0)      # raises loudly if there is no HIP device / no library
    yield ctx
    ctx.close()
