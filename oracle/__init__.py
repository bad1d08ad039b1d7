"""CPU oracle for the S-NeRF volumetric-render hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (torch fp32 on
the CPU for floating-point stages, numpy with an explicit accumulation order
for the index-producing samplers) of the reference algorithm for the path

    sample along ray -> encode -> tiny MLP (sigma, rgb) -> alpha composite

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``snerf_amd``) never imports it and fails loudly when the HIP library is
missing.

Parity pinning: every function here is checked against golden vectors that
``oracle/gen_golden.py`` captured by importing the reference's own Python
modules in the build container (the reference has no tests/fixtures of its
own for this path).  Fixtures live in ``tests/golden``.
The hash-grid encoder (reference CUDA source cannot be built or run here) is
the one stage whose parity is "unpinned by the reference"; it is pinned by
analytic known-answer tests instead (see ``oracle/grid.py``).

Modules
  common   formula weights, canonical (fp64-accumulate) sums for samplers
  classic  path B: render_rays / run_network / raw2outputs / sample_pdf
  mip      path A: MipNerfModel forward (warp sampling, IPE, 2 MLPs, composite)
"""
