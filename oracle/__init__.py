"""CPU oracle for the Robust e-NeRF volume-rendering hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a CPU restatement (pure PyTorch for the differentiable floating-point
pieces, plain C in ``oracle/csrc`` for the sequential ray-marching / packing
pieces) of the algorithm on the path named by ``BASELINE.json:north_star``.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it -- and there only as the checker, never as the thing
being measured or shipped.  The product package (``robust_e_nerf_amd``) never
imports it and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md section "Oracle"):

* Pieces whose source lives in ``/root/reference`` (MLP, SH encoder, space
  contraction, trunc_exp, ``rendering`` glue, ``pixel_params_to_ray``,
  ``LinearTrajectory`` + ``unitquat_slerp``, ``ContrastThreshold``,
  ``RefractoryPeriod``, ``Loss``, supervision-timestamp derivation) are PINNED:
  ``tests/golden/make_golden.py`` imports the reference's own Python (with
  third-party stubs) in the build container, runs it on seeded inputs and
  commits the input/output vectors under ``tests/golden/``; the oracle is
  checked against those vectors in ``tests/test_oracle_golden.py``.
* Pieces that live in third-party dependencies absent from ``/root/reference``
  -- nerfacc==0.3.1 (ray marching, visibility, packed transmittance, occupancy
  grid), tinycudann @ master (multi-resolution hash grid) and roma==1.2.7
  (quaternion algebra) -- are restated from their published algorithms and are
  PARITY UNPINNED by any reference test (the reference has no tests); they are
  anchored on the reference's call sites (shapes, dtypes, argument meaning).

Every function cites the reference ``file:line`` it follows.
"""
