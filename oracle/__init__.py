"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the evfly perception hot path.

This package is the parity oracle for `evfly_amd`. It restates, in numpy / plain C /
torch-CPU functional code, the algorithms of the reference files cited in each
function (paths relative to the reference repository root). It is NOT part of the
product: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import it, and only as the checker. `evfly_amd` never imports it and fails
loudly when the HIP library is missing.

Parity pin: the reference has no tests / golden vectors for this path (SURVEY.md
§4), so the oracle is pinned by fixtures generated in the build container by
importing the *reference itself* (`tests/golden/make_golden.py`, committed together
with its outputs `tests/golden/*.npz`). `tests/test_oracle_golden.py` checks every
oracle function against those fixtures. The two ROS C++ accumulator nodes cannot be
compiled here (ROS headers absent), so their 10-line loop bodies are restated in
`oracle/accum.c` and pinned only by hand-derived known answers ("parity unpinned"
for A3/A4 beyond those). `oracle/rectify.py` (Aligner.align, N3) restates two OpenCV
calls; OpenCV is absent from the reference tree and from this container: "parity
unpinned" for it as well. `oracle/sim.py` is pinned by running the reference's own
function body (G10).
"""
