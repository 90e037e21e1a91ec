"""CPU oracle for the NeO-360 ray-marching hot path.

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
anything from this package, and only as the checker / reported CPU baseline.
The product package (`neo360_amd`, sources in `neo-360_amd/`) never imports it
and fails loudly when its HIP library is missing.

What it is: a restatement, in plain fp32 PyTorch-on-CPU (fp64 NumPy for the
ray/AABB slab test), of the algorithm the reference executes on the path named
by BASELINE.json:north_star.  Each function cites the reference file:line it
follows (paths relative to the upstream repository root).  The reference is a
PyTorch program, so the restatement uses the same numerical substrate
(torch CPU kernels: cumsum/cumprod with double accumulators, Sleef sin, ...).

Pinning status: PINNED.  The reference ships no tests or golden vectors
(SURVEY.md §4), so the oracle is pinned against outputs of the reference
itself, produced in the build container by importing the reference's Python
under import stubs (`tests/golden/_ref_loader.py`) and committed as fixtures by
`tests/golden/make_golden.py` (inputs are regenerated from the build-owned
integer-hash generator `neo360_amd.synth`, fixtures hold expected outputs).
`tests/test_oracle_golden.py` checks every oracle stage and the end-to-end
renders against those fixtures on every CPU run; `tests/test_oracle_vs_reference.py`
re-derives them live when /root/reference is present.
"""
from . import rays, sampling, encoding, compositing, gather, mlp, vanilla, neo360, pixelnerf, pillar, training  # noqa: F401
