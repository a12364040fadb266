import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def bunny_scene_128():
    """The reference scene with bunnies only, 128x128 image, 16,384 ray slots."""
    from raytracer_amd import scene as S
    return S.build_scene([S.BUNNY, S.BUNNY], 128, 128, 128 * 128)


@pytest.fixture(scope="session")
def bunny_scene_64():
    from raytracer_amd import scene as S
    return S.build_scene([S.BUNNY, S.BUNNY], 64, 64, 8192)


@pytest.fixture(scope="session")
def standin_ply(tmp_path_factory):
    """A reduced dragon stand-in (about 70k triangles) for tests that need a second model."""
    from raytracer_amd import scene as S
    path = str(tmp_path_factory.mktemp("models") / "standin_small.ply")
    return S.write_standin_ply(path, 264, 132)


@pytest.fixture(scope="session")
def standin_full_ply(tmp_path_factory):
    """The full-size dragon stand-in: 870,488 triangles, the hierarchy shape of dragon_vrip.ply
    (model root of 64 boxes, 64 inner groups of 128, 8,192 leaves)."""
    from raytracer_amd import scene as S
    path = str(tmp_path_factory.mktemp("models") / "standin_full.ply")
    return S.write_standin_ply(path, 934, 466)
