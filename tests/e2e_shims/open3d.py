"""`import open3d` stand-in (train.py:14, utils/vis_utils.py:4): only utils.vis_utils.save_points touches it, and train.py never
calls that.  The attribute chain exists so that the import succeeds; using it raises."""


class _Missing:
    def __init__(self, name):
        self._name = name

    def __getattr__(self, k):
        return _Missing(self._name + "." + k)

    def __call__(self, *a, **k):
        raise RuntimeError("%s: open3d is not installed in this image (tests/e2e_shims stand-in)" % self._name)


geometry = _Missing("open3d.geometry")
utility = _Missing("open3d.utility")
io = _Missing("open3d.io")
