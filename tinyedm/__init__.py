"""Alias package: makes the reference's import / ``_target_`` paths (``tinyedm.EDM``,
``tinyedm.networks.Denoiser``, ``tinyedm.edm.EDM`` ...) resolve to the MI355X-native implementation."""
import sys

import tinyedm_amd as _impl
from tinyedm_amd import *  # noqa: F401,F403
from tinyedm_amd import config, edm, ema, metric, networks, solvers, utils  # noqa: F401

# (`generate` is a real shim module, tinyedm/generate.py, so that `python -m tinyedm.generate` runs)
for _name in ("config", "edm", "ema", "metric", "networks", "solvers", "utils", "trainer", "datamodules", "callbacks",
              "graph"):
    try:
        _mod = __import__(f"tinyedm_amd.{_name}", fromlist=["_"])
    except ImportError:
        continue
    sys.modules[f"tinyedm.{_name}"] = _mod
    globals()[_name] = _mod
__all__ = _impl.__all__
