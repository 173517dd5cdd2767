import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'frame-interpolation_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


def oracle_options(opt):
    from oracle import film_oracle as fo
    return fo.Options(pyramid_levels=opt.pyramid_levels, fusion_pyramid_levels=opt.fusion_pyramid_levels,
                      specialized_levels=opt.specialized_levels, sub_levels=opt.sub_levels,
                      flow_convs=tuple(opt.flow_convs), flow_filters=tuple(opt.flow_filters), filters=opt.filters)


@pytest.fixture(scope='session')
def tiny_weights():
    from film_hip import weights as W, options as O
    return W.make_synthetic_weights(O.TINY, seed=0)
