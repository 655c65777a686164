"""CPU oracle for the magnify marker-detection hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy/C restatement of the
reference's algorithm (FordyceLab/magnify v0.12.5) for the hot path named in
BASELINE.json.  It may be imported only by ``tests/``, by
``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of ``bench.py`` --
always as the *checker* / reported baseline, never as the thing measured or
shipped.  The product (``magnify_amd``) never imports it and fails loudly when
its HIP extension is missing.

Pinning (see DESIGN.md "Oracle"):
  * ``ref_numeric``  -- pinned bit-for-bit by ``tests/golden/*.npz``, which were
    generated in the build container by executing the reference's own
    ``src/magnify/utils.py`` / ``find.py`` (``tests/golden/make_golden.py``).
  * ``ref_opencv``   -- restates OpenCV 4.13 semantics (GaussianBlur, Scharr,
    Canny, circle).  OpenCV is an un-vendored third-party dependency that is not
    installed here: PARITY UNPINNED at bit level for these four functions;
    pinned only end-to-end through the reference's own tolerance tests.
  * ``ref_contours`` -- Suzuki-Abe border following + CHAIN_APPROX_SIMPLE + arcLength as ``cv.findContours`` /
    ``cv.arcLength`` implement them (filter.py:51-52): PARITY UNPINNED against OpenCV itself; independent of the
    product's Moore tracing, which it checks.
  * ``ref_pipeline`` -- restates xarray/dask-level code that cannot run here
    (stitch, flatfield_correct, BeadFinder, ButtonFinder); stitch is pinned by
    the reference's exact index assertions (tests/test_stitch.py), the finders
    by the reference's tolerance tests, flatfield_correct and the ROI-reduce
    expressions have no reference test: PARITY UNPINNED for those two.
"""
