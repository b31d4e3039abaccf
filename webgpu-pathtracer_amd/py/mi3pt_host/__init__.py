"""Host-side Python mirror of the reference's Renderer / Pass / Scene API over
libmi3pt.so (the MI355X back-end).  See renderer.py."""
from . import capi, layout, scenes  # noqa: F401
from .renderer import (AccumulatePass, FullscreenPass, Pass, RaytracePass, RaytracingCamera,  # noqa: F401
                       RaytracingScene, Renderer, RollingAverage)
