"""CPU oracle for the OAI_analysis_2 per-volume dense path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product path (``oai_analysis_2_amd``) never imports this package and fails
loudly when the HIP library is missing.

What is restated here (torch-CPU / numpy, fp32 like the reference's CPU path):

* ``oracle.seg``      -- overlap-tile partition, 3D U-Net forward, sigmoid /
                         threshold, stitch (reference: oai_analysis/segmentation/
                         {image_transforms,networks,segmenter}.py).  PINNED: checked
                         against an import of the reference itself with seeded
                         weights (tests/golden/make_golden.py -> tests/golden/*.npz).
* ``oracle.icon``     -- ICON gradICON registration forward (third-party
                         icon_registration==1.1.2, pinned in the reference's
                         pyproject.toml:35 but absent from /root/reference and from
                         this image).  PARITY UNPINNED: restated from the package's
                         published structure, anchored on the reference call sites
                         oai_analysis/registration.py:20,25.
* ``oracle.resample`` -- prob-map resample through phi (ITK ResampleImageFilter +
                         DisplacementFieldTransform, call sites test/test_all.py:42-52,
                         oai_analysis/dask_processing.py:95-111).  PARITY UNPINNED
                         (ITK is absent; the reference's asserts there are commented out).
"""
