import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'frame-interpolation_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def _keep_large_allocations_in_the_heap():
    """The oracle's numpy temporaries are 100s of MB each; glibc serves such requests with mmap / munmap, i.e. fresh zeroed
    pages and their page faults on every one (on this VM, which hands freed memory back to its host: 3 minutes of system
    time in ONE 1024x768 oracle forward).  Without mmap-served requests and without heap trimming the freed pages are reused."""
    try:
        import ctypes
        libc = ctypes.CDLL('libc.so.6')
        M_TRIM_THRESHOLD, M_MMAP_MAX = -1, -4
        libc.mallopt(M_MMAP_MAX, 0)                       # no mmap-served requests at all: everything comes from the heap ...
        libc.mallopt(M_TRIM_THRESHOLD, (1 << 31) - 1)     # ... and freed heap pages are kept (and reused) instead of returned
    except Exception:   # pragma: no cover - a speed-up only
        pass


_keep_large_allocations_in_the_heap()


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')
    # FILM_TEST_EXECUTOR=0|1|2: every engine a test creates starts with that value of option "graph" (1 = the hipGraph replay that was
    # the default until round 5) - for re-running the suite on another executor, e.g. to reproduce profiles/r05_hipgraph_first_launch_crash.md
    forced = os.environ.get('FILM_TEST_EXECUTOR')
    if forced:
        from film_hip.engine import FilmEngine
        orig = FilmEngine.__init__

        def init(self, *a, **k):
            orig(self, *a, **k)
            if self.device >= 0:
                self.set_option('graph', int(forced))
        FilmEngine.__init__ = init


def has_extra_families() -> bool:
    """The loaded libfilm_hip flavour holds the opt-in kernel families (FILM_EXTRA_FAMILIES=1 build: bf16 precision modes,
    F(2,3) / halo kernels).  The default library - what the driver builds and tests - does not."""
    from film_hip.engine import FilmEngine
    try:
        return FilmEngine.has_extra_families()
    except Exception:   # library not built: the tests that need it fail on their own
        return False


needs_extra_families = pytest.mark.skipif(
    "not __import__('conftest').has_extra_families()",
    reason='needs the FILM_EXTRA_FAMILIES=1 flavour of the library (FILM_EXTRA_FAMILIES=1 python -m film_hip.build; run pytest with FILM_EXTRA_FAMILIES=1)')


def oracle_options(opt):
    from oracle import film_oracle as fo
    return fo.Options(pyramid_levels=opt.pyramid_levels, fusion_pyramid_levels=opt.fusion_pyramid_levels,
                      specialized_levels=opt.specialized_levels, sub_levels=opt.sub_levels,
                      flow_convs=tuple(opt.flow_convs), flow_filters=tuple(opt.flow_filters), filters=opt.filters)


@pytest.fixture(scope='session')
def tiny_weights():
    from film_hip import weights as W, options as O
    return W.make_synthetic_weights(O.TINY, seed=0)
