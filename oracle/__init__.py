"""CPU oracle for the SVD denoise hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU fp32 restatement of the algorithm behind
``UNetSpatioTemporalConditionModel.forward`` / ``ControlNetModel.forward`` /
``EulerDiscreteScheduler`` as used by the reference pipelines.  It exists so the
HIP path can be checked; it is never shipped and never measured as the product.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``this_and_that_vdm_amd/`` does.

PARITY STATUS (read before trusting it):
  * composition code (block wiring, eps quirks, residual placement, the
    ``time_context`` reshuffle, ControlNet zero-convs, the denoise loop) is PINNED:
    ``tests/golden/make_golden.py`` imports the reference's own
    ``svd/*.py`` in the build container and the fixtures under ``tests/golden``
    were produced by that code;
  * leaf arithmetic (ResnetBlock2D, Attention, GEGLU, AlphaBlender, Timesteps,
    EulerDiscreteScheduler ...) lives in the un-vendored third-party package
    ``diffusers==0.25.1`` (reference ``requirements.txt:23``) whose source is not
    in ``/root/reference``; it is restated here from its published algorithm and
    is therefore **parity unpinned** at the leaf level (no diffusers wheel, no
    network).  Known-answer identities in ``tests/test_oracle_*.py`` (parameter
    counts, zero-ControlNet == VL, sigma table) are the available anchors.
"""
