"""Hyper-parameters of the film_net interpolator.

Mirrors models/film_net/options.py:20-80 of the reference (class ``Options``); the
defaults here are the *published* architecture from
training/config/film_net-L1.gin:17-23 (the VGG and Style gins use the same net), because
that is the only configuration the released SavedModels exist for.
"""
from __future__ import annotations

import dataclasses
from typing import Sequence, Tuple


@dataclasses.dataclass(frozen=True)
class Options:
    pyramid_levels: int = 7
    fusion_pyramid_levels: int = 5
    specialized_levels: int = 3
    sub_levels: int = 4
    flow_convs: Tuple[int, ...] = (3, 3, 3, 3)
    flow_filters: Tuple[int, ...] = (32, 64, 128, 256)
    filters: int = 64

    def validate(self) -> None:
        """Constraints of the reference plus the channel-alignment rules of the HIP engine."""
        if self.pyramid_levels < self.fusion_pyramid_levels:
            # models/film_net/interpolator.py:120-122
            raise ValueError('config.pyramid_levels must be greater than or equal to '
                             'config.fusion_pyramid_levels.')
        if not (1 <= self.specialized_levels <= self.pyramid_levels):
            raise ValueError('specialized_levels must be in [1, pyramid_levels]')
        if len(self.flow_convs) != self.specialized_levels + 1 or \
                len(self.flow_filters) != self.specialized_levels + 1:
            raise ValueError('flow_convs / flow_filters need specialized_levels+1 entries')
        if self.sub_levels < 1 or self.sub_levels > self.specialized_levels + 1:
            raise ValueError('sub_levels must be within [1, specialized_levels+1]')
        if self.filters <= 0 or self.filters % 32:
            raise ValueError('HIP engine: filters must be a positive multiple of 32')
        if any(f != 32 and f % 64 for f in self.flow_filters):
            raise ValueError('HIP engine: flow_filters must be 32 or multiples of 64')

    @property
    def align(self) -> int:
        """Input H, W must be divisible by this (options.py:36-37)."""
        return 2 ** (self.pyramid_levels - 1)


PUBLISHED = Options()

# A small architecture with the same topology; used by fast tests only.
TINY = Options(pyramid_levels=4, fusion_pyramid_levels=3, specialized_levels=2, sub_levels=3,
               flow_convs=(2, 2, 2), flow_filters=(32, 64, 64), filters=32)
