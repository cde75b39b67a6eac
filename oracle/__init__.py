"""CPU oracle for the dynamic-video-depth test-time-optimisation hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a plain PyTorch-CPU restatement of the reference algorithm
(google/dynamic-video-depth, /root/reference) for the path named in
BASELINE.json: unproject -> scene-flow MLP (Euler multi-step) -> flow-warped
reprojection -> masked losses -> backward -> acceleration regulariser -> Adam.
The reference itself is 100 % PyTorch, so the restatement uses the same ATen
CPU operators in the same order; that makes it bit-identical to the reference
on the same host (checked by tests/test_oracle_vs_golden.py against fixtures
generated from the *real* reference by tests/golden/make_golden.py).

Only these may import it: tests/, __graft_entry__.smoke(), and the
`cpu_baseline` leg of bench.py.  The product package (dvd_hip) never imports
it and has no CPU fallback: it raises if the HIP library is missing.

Parity pinning status: the reference ships no tests or golden vectors
(SURVEY.md section 4), so the pins are the fixtures under tests/golden/ that were
produced by importing /root/reference in the build container (script
committed next to them).
"""

from . import geometry, sceneflow_mlp, losses  # noqa: F401
