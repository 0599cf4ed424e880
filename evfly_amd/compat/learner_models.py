"""Shim: `import learner_models` -> evfly_amd.learner_models (see README.md in this directory)."""
import os as _os, sys as _sys
_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.append(_root)
from evfly_amd.learner_models import *  # noqa: F401,F403,E402
