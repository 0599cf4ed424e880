"""Shim: `import dataloading` -> evfly_amd.dataloading (see README.md in this directory)."""
import os as _os, sys as _sys
_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.append(_root)
from evfly_amd.dataloading import *  # noqa: F401,F403,E402
from evfly_amd.dataloading import dataloader, preload, find_unmatched_indices  # noqa: F401,E402
