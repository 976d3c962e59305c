import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_sessionstart(session):
    """The HIP libraries are build artefacts (git-ignored).  build() compiles them when they are missing
    OR older than any of their sources (hipcc cross-compiles gfx950 without a GPU, ~20 s; a no-op when
    fresh), so the suite can never pass against a stale binary after an edit to csrc/ or include/."""
    from proteus_amd import build
    build.build()
    build.build_lab()
