"""MI355X-native Hourglass (encoder + critic head + decoder/mask head) training and inference path of
ndrwmlnk/critic-guided-segmentation-of-rewarding-objects-in-first-person-views.

Python here is host-side plumbing (tensors, streams, checkpoints, CLI); the arithmetic runs in the
hand-written gfx950 kernels of ``libcgs_hip.so`` (C ABI: include/cgs_hip.h).  The directory name is not a
Python identifier; import it through the ``cgs_amd`` alias module at the repository root."""
from . import _lib, spec, hourglass, nets, engine, handler, cli, parallel, dataformat  # noqa: F401
from .nets import NewCritic, UnetDecoder  # noqa: F401
from .engine import HourglassEngine  # noqa: F401

__all__ = ["_lib", "spec", "hourglass", "nets", "engine", "handler", "cli", "parallel", "dataformat", "NewCritic", "UnetDecoder", "HourglassEngine"]
