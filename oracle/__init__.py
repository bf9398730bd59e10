"""CPU oracle for the SR generator hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the reported CPU baseline.
The product path (``super-resolution_amd/srmi``) never imports this package.

Pinning status (see DESIGN.md "Oracle"):
  * SRResNet/SRGAN Generator, Discriminator, TVLoss, MeanShift: pinned against
    outputs of the reference itself, imported in the build container by
    ``oracle/make_golden.py``; vectors committed under ``tests/golden/``.
  * VGG19 perceptual features (torchvision weights), Y-PSNR/SSIM (scikit-image),
    SRCNN / ESRGAN-RRDB (Keras scripts): parity unpinned - the third-party code
    and weights are absent from the container; restated from the reference's
    call sites only.
"""
