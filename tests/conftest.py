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
    F(2,3) / halo kernels).  The default library - what a default plan needs - does not."""
    from film_hip.engine import FilmEngine
    try:
        return FilmEngine.has_extra_families()
    except Exception:   # library not built: the tests that need it fail on their own
        return False


_EXTRA_LIB = [None]


@pytest.fixture
def extra_families_library():
    """Round-5 verdict: ONE `pytest -m gpu` run must cover every kernel that ships.  __graft_entry__.build() makes both flavours of the
    library; a test that needs the opt-in families (precision modes, F(2,3) / halo kernels) runs with film_hip/libfilm_hip_extra.so
    bound for every engine it creates (same sources + those families; the two libraries coexist in the process), whatever flavour
    the rest of the run uses.  Skips only when that file has not been built."""
    from film_hip import engine
    if has_extra_families():          # a FILM_EXTRA_FAMILIES=1 run: already the flavour
        yield
        return
    path = os.path.join(PKG, 'film_hip', 'libfilm_hip_extra.so')
    if not os.path.isfile(path):
        pytest.skip('film_hip/libfilm_hip_extra.so has not been built (python __graft_entry__.py builds both flavours)')
    if _EXTRA_LIB[0] is None:
        _EXTRA_LIB[0] = engine.load_library(path)
    saved = engine._lib
    engine._lib = _EXTRA_LIB[0]
    try:
        assert has_extra_families()
        yield
    finally:
        engine._lib = saved


needs_extra_families = pytest.mark.usefixtures('extra_families_library')


@pytest.fixture
def prefer_extra_families_library():
    """For tests that cover MORE with the opt-in families but also run without them (`if has_extra_families(): ...`): binds
    libfilm_hip_extra.so when it has been built, the default library otherwise."""
    from film_hip import engine
    path = os.path.join(PKG, 'film_hip', 'libfilm_hip_extra.so')
    if has_extra_families() or not os.path.isfile(path):
        yield
        return
    if _EXTRA_LIB[0] is None:
        _EXTRA_LIB[0] = engine.load_library(path)
    saved = engine._lib
    engine._lib = _EXTRA_LIB[0]
    try:
        yield
    finally:
        engine._lib = saved


prefers_extra_families = pytest.mark.usefixtures('prefer_extra_families_library')


def oracle_options(opt):
    from oracle import film_oracle as fo
    return fo.Options(pyramid_levels=opt.pyramid_levels, fusion_pyramid_levels=opt.fusion_pyramid_levels,
                      specialized_levels=opt.specialized_levels, sub_levels=opt.sub_levels,
                      flow_convs=tuple(opt.flow_convs), flow_filters=tuple(opt.flow_filters), filters=opt.filters)


@pytest.fixture(scope='session')
def tiny_weights():
    from film_hip import weights as W, options as O
    return W.make_synthetic_weights(O.TINY, seed=0)
