"""TEST INFRASTRUCTURE -- CPU restatement of the simulator pilot's event estimate.

Follows envtest/ros/run_competition.py:603-635 (`AgilePilotNode.compute_events`) with SMALL_EPS = 1e-5 (:29).
Pinned by tests/golden/g10_difflog.npz, which tests/golden/make_golden.py produces by executing the reference's
own function body (extracted from the file with `ast`, because the module itself imports rospy / cv_bridge).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import numpy as np

SMALL_EPS = 1e-5


def compute_events(im, prev_im, neg_thresh=0.2, pos_thresh=0.2):
    """im, prev_im: (H, W) arrays as im_callback stores them (float32 / 255; the very first prev_im is the float64
    zeros of :341). Returns the `self.events` array of :624-633 (dtype of difflog)."""
    difflog = np.log(im + SMALL_EPS) - np.log(prev_im + SMALL_EPS)                  # :621
    events = np.zeros_like(difflog)                                                 # :624
    if np.abs(difflog).max() < max(pos_thresh, neg_thresh):                         # :626-627
        return events
    pos = difflog > 0.0                                                             # :630-633
    neg = difflog < 0.0
    events[pos] = (difflog[pos] // pos_thresh) * pos_thresh
    events[neg] = (difflog[neg] // -neg_thresh) * -neg_thresh
    return events


def difflog(im, prev_im):
    """The intermediate of :621, for the boundary analysis in the parity tests."""
    return np.log(im + SMALL_EPS) - np.log(prev_im + SMALL_EPS)


def command_velocity(x, desiredVel, pos_x):
    """:577-585: scale the unit command and apply the hard-coded acceleration ramp."""
    v = np.asarray(x, dtype=np.float64) * desiredVel
    if pos_x < 2.0:
        v[0] = max(1.0, (pos_x / 2.0) * desiredVel)
    return v
