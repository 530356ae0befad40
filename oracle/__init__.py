"""oracle/ — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Plain torch-CPU / numpy fp32 restatement (our own code, `[N, ...]` layouts) of the
kyungminn/PBHC humanoidverse motion-tracking hot path: quaternion algebra, MJCF skeleton
tables, motion-library FK + filtered velocities, phase lookup (lerp/slerp), the
`LeggedRobotMotionTracking` step (body extension, tracking diffs, termination, reward terms,
reset, observation assembly, history) and the MHPPO maths (GAE, advantage normalisation,
surrogate / value / entropy losses).  Every function cites the reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package, and only as the checker / the timed CPU baseline — never as the product path.  The
product (`pbhc_amd`) runs hand-written HIP kernels and raises if its extension is missing.

Parity pinning: the reference has no tests or golden vectors for this path.  The
oracle is pinned against outputs of the reference itself, run in the build container by
`oracle/ref_harness/gen_golden.py` (unmodified reference code from /root/reference, stand-ins
only for absent non-arithmetic third-party modules) and committed under `tests/golden/`.
"""
