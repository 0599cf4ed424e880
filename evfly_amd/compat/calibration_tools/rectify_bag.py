"""Bare-name shim: `from calibration_tools.rectify_bag import Aligner` (evfly_ros/run.py:25) resolves here when
`evfly_amd/compat` is on sys.path."""
from evfly_amd.calibration_tools.rectify_bag import *  # noqa: F401,F403
from evfly_amd.calibration_tools.rectify_bag import Aligner, Camera, CameraSystem, remap_img  # noqa: F401
