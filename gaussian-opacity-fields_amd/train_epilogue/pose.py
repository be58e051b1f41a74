"""What train.py does with the camera pose every iteration (train.py:177-179):

    c2w = (viewpoint_cam.world_view_transform.T).inverse()
    normal2 = c2w[:3, :3] @ render_normal.reshape(3, -1)

`.inverse()` of a 4x4 device tensor is five rocSOLVER launches, three copies and a host synchronisation (measured on MI355X,
profiles/r05_full_loop_kernel_stats.md: 25 us of kernels + 83 us of idle GPU per iteration) for a matrix that never changes; the
3 x 3 x N product is handed to a GEMM library (128 us forward + 125 us backward at 1600x1063 for 40 MB of streaming).  Neither line
can be edited -- train.py runs unchanged -- but both start from an attribute of the Camera object, so the launcher
(launch/run_reference_script.py) hands the script a pose matrix that REMEMBERS its inverse and whose 3x3 block applies itself with
one streaming launch (include/gof_train_hip.h: gof_rot3_apply).  Values: the inverse is torch.linalg.inv's, computed once per
camera (and again if the matrix is modified in place); the product is a . x with two fused multiply-adds where the GEMM's
summation order is its own -- the last bit of a float32 dot product of three terms.

Everything else done with these tensors behaves as on a plain torch.Tensor (results are plain tensors)."""
import torch

from . import _backend as B

_MIN_COLUMNS = 4096          # below this the GEMM library's launch is as good as ours


def _plain(t):
    return t.as_subclass(torch.Tensor)


class _PlainResults(torch.Tensor):
    """A tensor subclass whose operations give PLAIN tensors (the subclass carries behaviour for two spellings only, it must not
    spread through a script's arithmetic)."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))


class _Rot3Apply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m33, x):
        ctx.save_for_backward(m33)
        return B.rot3_apply(m33, x, transpose=False)

    @staticmethod
    def backward(ctx, g):
        (m33,) = ctx.saved_tensors
        return None, B.rot3_apply(m33, g, transpose=True)


class SmallMatrix(_PlainResults):
    """The cached inverse pose and its sub-blocks: `M[:3, :3] @ X` with X [3, N] on the device runs gof_rot3_apply."""

    def __getitem__(self, idx):
        r = torch.Tensor.__getitem__(_plain(self), idx)
        return r.as_subclass(SmallMatrix) if isinstance(r, torch.Tensor) and r.dim() == 2 else r

    def __matmul__(self, other):
        if not isinstance(other, torch.Tensor) and hasattr(type(other), "_gof_rmatmul"):      # a deferred intermediate of train.py:175-178 (deferred.py)
            return other._gof_rmatmul(self)
        m = _plain(self)
        if (isinstance(other, torch.Tensor) and m.dim() == 2 and tuple(m.shape) == (3, 3) and other.dim() == 2 and other.shape[0] == 3
                and other.shape[1] >= _MIN_COLUMNS and m.device.type == "cuda" and other.device == m.device
                and m.dtype == torch.float32 and other.dtype == torch.float32 and not m.requires_grad):
            return _Rot3Apply.apply(m, other)
        return torch.matmul(m, other)


class _PoseTransposed(_PlainResults):
    """`world_view_transform.T`: a transposed view that knows whose transpose it is."""
    _gof_parent = None

    def inverse(self):
        p = self._gof_parent
        if p is None:
            return torch.linalg.inv(_plain(self))
        cache = p.__dict__.get("_gof_inverse_of_T")
        if cache is None or cache[0] != p._version:
            inv = torch.linalg.inv(_plain(self)).as_subclass(SmallMatrix)      # (once per camera: the only synchronising call left)
            cache = p.__dict__["_gof_inverse_of_T"] = (p._version, inv)
        return cache[1]


class PoseMatrix(_PlainResults):
    """Camera.world_view_transform as the launcher hands it to the script: a plain float32 [4,4] tensor in every respect but one --
    `.T.inverse()` is computed once."""

    @staticmethod
    def wrap(t):
        if isinstance(t, PoseMatrix) or not isinstance(t, torch.Tensor) or t.dim() != 2:
            return t
        return t.as_subclass(PoseMatrix)

    @property
    def T(self):
        t = _plain(self).T.as_subclass(_PoseTransposed)
        t._gof_parent = self
        return t
