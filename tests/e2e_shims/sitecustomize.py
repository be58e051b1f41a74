"""Environment compatibility for the end-to-end runs of the UNCHANGED reference (test infrastructure; active only when this
directory is on PYTHONPATH): scene/dataset_readers.py:215 builds its RGB image with `Image.fromarray(int8 array, "RGB")`, which the
Pillow releases of the reference's environment (9.x) accepted -- an explicit mode made them take the buffer as raw bytes -- and the
Pillow in this image rejects.  The wrapper restores exactly that: a signed-byte array passed WITH a mode is viewed as unsigned bytes."""
try:
    import numpy as _np
    import PIL.Image as _Image

    _fromarray = _Image.fromarray

    def fromarray(obj, mode=None):
        if mode is not None and isinstance(obj, _np.ndarray) and obj.dtype == _np.int8:
            obj = obj.view(_np.uint8)
        return _fromarray(obj, mode)

    _Image.fromarray = fromarray
except Exception:      # pragma: no cover  (never break interpreter start-up)
    pass

# GOF_E2E_SEED=<n> (tests/test_trajectory_gpu.py): the reference's train.py leaves `safe_state(args.quiet)` commented out (train.py:365),
# so Python's `random` -- the camera order, train.py:103 -- starts from OS entropy; two runs that are to be COMPARED step by step
# need the same sequence.  Seeds `random` and numpy's global generator at interpreter start (torch's default generators start from
# a fixed seed on their own).
try:
    import os as _os
    if _os.environ.get("GOF_E2E_SEED"):
        import random as _random
        _random.seed(int(_os.environ["GOF_E2E_SEED"]))
        try:
            import numpy as _np2
            _np2.random.seed(int(_os.environ["GOF_E2E_SEED"]))
        except Exception:
            pass
except Exception:      # pragma: no cover
    pass
