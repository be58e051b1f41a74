"""train.py:150-189 evaluated by the one-call loss (fused_loss.training_loss, gof_train_loss) WITHOUT editing train.py.

train.py composes its loss inline from the rasterizer's image: l1_loss, ssim, a mean, depth_to_normal, F.normalize, a 3x3 matmul, a
product, a sum, two more means and six scalar operations -- ~45 launches and their autograd even with every helper on HIP (0.35 ms
of a 3.4 ms iteration at 1600x1063).  None of these lines can be edited, but every one of them starts from something this backend
hands the script: the image (diff_gaussian_rasterization.RenderedImage), the three helpers the launcher rebinds by name (l1_loss,
ssim, depth_to_normal) and the camera's pose (pose.PoseMatrix).  With ``enable()`` (launch/run_reference_script.py; GOF_EAGER_LOSS=1
keeps the eager mirrors) those hand back DEFERRED values:

    image's channel slices      real tensors that remember their channels (RenderedChannels)
    l1_loss(image, gt), ssim(image, gt), rendering[8].mean(), (1 - (world_normal * depth_normal).sum(dim=0)).mean()
                                DeferredLoss: a linear combination  sum_i a_i term_i + c  of the four terms of train.py:156-182;
                                python-number arithmetic (+, -, *, /) only changes a_i and c -- no launch
    depth_to_normal(view, depth[None]), F.normalize(rendering[3:6], p=2, dim=0), .permute / .reshape / c2w[:3,:3] @ . / * / .sum(dim=0) / 1 - .
                                DeferredTensor: the name of an intermediate of train.py:170-181, nothing computed
    loss.backward()             the combination's coefficients give (lambda_dssim, lambda_depth_normal, lambda_distortion) and a
                                scale; ONE training_loss call (five launches) and its backward deliver d loss / d rendering to
                                the rasterizer's backward; .item() afterwards reads the terms that call left

    + / - a 0-dim tensor        carried along as it is (its own autograd graph): the decoupled-appearance L1 of train.py:158-159 -- the
                                reference's TNT / DTU runs -- goes through a network and stays torch code; `0.8 * that + 0.2 * (1 - ssim)`
                                keeps the rest deferred, and loss.backward() starts ONE pass from the fused gradient and from the tensor

ANY other use of a deferred object -- an operator, torch function, attribute or argument pattern not listed above, a different
ground-truth tensor for ssim than for l1_loss, a tensor-valued factor -- makes it compute itself EAGERLY with exactly the mirrors
the launcher bound before this module existed (loss_utils.l1_loss / ssim, depth_utils.depth_to_normal, torch for the rest) and
continue as the plain tensor: a script that composes its loss differently runs as before, only without (that part of) the saving.  Values: training_loss's (tests: the trajectory and epilogue tests of
tests/test_e2e_scripts_gpu.py and tests/test_train_epilogue_gpu.py hold it against the eager composition and the oracle)."""
import numbers
import operator

import torch

from . import depth_utils, fused_loss, loss_utils

_ENABLED = False
stats = {"fused_backwards": 0, "eager_terms": 0, "eager_tensors": 0}
# what the deferred objects call when they have to compute something (tests replace entries to run the logic without a GPU)
impl = {"l1": loss_utils.l1_loss, "ssim": loss_utils.ssim, "depth_to_normal": depth_utils.depth_to_normal, "fused": fused_loss.training_loss_gradient}


def _plain(t):
    with torch._C.DisableTorchFunctionSubclass():
        return t.as_subclass(torch.Tensor)


class _Frame:
    """One rasterizer forward: the image and what the script has said about its loss so far."""
    __slots__ = ("image", "gt", "view", "c2w33", "dn_err", "terms", "terms_host", "_eager")

    def __init__(self, image):
        self.image, self.gt, self.view, self.c2w33, self.dn_err, self.terms, self.terms_host = image, None, None, None, None, None, None
        self._eager = {}            # what had to be computed eagerly for this frame (slices, terms): non-empty = the frame's loss stays eager

    @staticmethod
    def of(image):
        f = image.__dict__.get("_gof_frame")
        if f is None:
            f = image.__dict__["_gof_frame"] = _Frame(image)
        return f

    def accept_gt(self, image_slice, gt):
        if not (isinstance(gt, torch.Tensor) and type(gt) is torch.Tensor and gt.shape == image_slice.shape and gt.dtype == torch.float32
                and gt.device == image_slice.device and not gt.requires_grad):
            return False
        if self.gt is None:
            self.gt = gt
        return self.gt is gt

    def channels(self, lo, hi, squeeze):
        """the eager slice (through the image's own slicing: its gradient reaches the rasterizer's backward as before)"""
        key = ("ch", lo, hi, squeeze)
        if key not in self._eager:
            self._eager[key] = _plain(self.image[lo] if squeeze else self.image[lo:hi])
        return self._eager[key]

    def eager_term(self, name):
        if name not in self._eager:
            stats["eager_terms"] += 1
            if name == "l1":
                v = impl["l1"](self.channels(0, 3, False), self.gt)
            elif name == "ssim":
                v = impl["ssim"](self.channels(0, 3, False), self.gt)
            elif name == "dist":
                v = self.channels(8, 9, True).mean()
            else:
                v = self.dn_err.eager().mean()
            self._eager[name] = v
        return self._eager[name]


def _is_number(x):
    return isinstance(x, numbers.Real) and not isinstance(x, bool)


def _materialize(x):
    if isinstance(x, (DeferredLoss, DeferredTensor)):
        return x.eager()
    if isinstance(x, (tuple, list)):
        return type(x)(_materialize(v) for v in x)
    if isinstance(x, dict):
        return {k: _materialize(v) for k, v in x.items()}
    return x


class _Deferred:
    """what the two deferred kinds share: every use nobody listed computes the value and goes on with the plain tensor"""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return func(*_materialize(args), **_materialize(kwargs or {}))

    def __getattr__(self, name):                        # only attributes the class does not define
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.tensor(), name)


def _binary(op, reflected=False):
    def f(self, other):
        a, b = self.tensor(), _materialize(other)
        return op(b, a) if reflected else op(a, b)
    return f


for _name, _op in (("add", operator.add), ("sub", operator.sub), ("mul", operator.mul), ("truediv", operator.truediv), ("matmul", operator.matmul),
                   ("pow", operator.pow), ("lt", operator.lt), ("le", operator.le), ("gt", operator.gt), ("ge", operator.ge)):
    setattr(_Deferred, "__%s__" % _name, _binary(_op))
    if _name in ("add", "sub", "mul", "truediv", "matmul", "pow"):
        setattr(_Deferred, "__r%s__" % _name, _binary(_op, True))
_Deferred.__neg__ = lambda self: -self.tensor()
_Deferred.__getitem__ = lambda self, idx: self.tensor()[idx]
_Deferred.__float__ = lambda self: float(self.tensor())
_Deferred.__bool__ = lambda self: bool(self.tensor())
_Deferred.__repr__ = lambda self: "deferred(%r)" % (self.tensor(),)
_Deferred.__format__ = lambda self, spec: format(self.tensor(), spec)


class DeferredTensor(_Deferred):
    """An intermediate of train.py:170-181 by NAME (kind); `parents` are what the eager computation would start from."""
    def __init__(self, frame, kind, shape, thunk):
        self.frame, self.kind, self.shape, self._thunk, self._value = frame, kind, torch.Size(shape), thunk, None

    def eager(self):
        if self._value is None:
            stats["eager_tensors"] += 1
            self._value = self._thunk()
        return self._value

    tensor = eager

    def dim(self):
        return len(self.shape)

    ndim = property(dim)

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def permute(self, *dims):
        dims = tuple(dims[0]) if len(dims) == 1 and isinstance(dims[0], (tuple, list)) else dims
        if self.kind == "dn_hw3" and dims == (2, 0, 1):
            return DeferredTensor(self.frame, "dn_chw", (3,) + tuple(self.shape[:2]), lambda: self.eager().permute(2, 0, 1))
        return self.eager().permute(*dims)

    def reshape(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape
        if self.kind == "unit" and shape == (3, -1):
            return DeferredTensor(self.frame, "unit_flat", (3, self.shape[1] * self.shape[2]), lambda: self.eager().reshape(3, -1))
        if self.kind == "world_flat" and shape == (3,) + tuple(self.frame.image.shape[1:]):
            return DeferredTensor(self.frame, "world", shape, lambda: self.eager().reshape(shape))
        return self.eager().reshape(*shape)

    def _times(self, other):
        if isinstance(other, DeferredTensor) and other.frame is self.frame and {self.kind, other.kind} == {"world", "dn_chw"}:
            a, b = self, other
            return DeferredTensor(self.frame, "prod", self.shape, lambda: a.eager() * b.eager())
        return NotImplemented

    def __mul__(self, other):
        r = self._times(other)
        return self.eager() * _materialize(other) if r is NotImplemented else r

    def __rmul__(self, other):
        r = self._times(other)
        return _materialize(other) * self.eager() if r is NotImplemented else r

    def sum(self, *args, **kwargs):
        if self.kind == "prod" and ((args == (0,) and not kwargs) or (not args and kwargs == {"dim": 0})):
            return DeferredTensor(self.frame, "dot", self.shape[1:], lambda: self.eager().sum(dim=0))
        return self.eager().sum(*args, **kwargs)

    def __rsub__(self, other):
        if self.kind == "dot" and _is_number(other) and other == 1:
            t = DeferredTensor(self.frame, "err", self.shape, lambda: 1 - self.eager())
            return t
        return _materialize(other) - self.eager()

    def mean(self, *args, **kwargs):
        if self.kind == "err" and not args and not kwargs and self.frame.dn_err is None:
            self.frame.dn_err = self
            return DeferredLoss(self.frame, {"dn": 1.0}, 0.0)
        return self.eager().mean(*args, **kwargs)

    def _gof_rmatmul(self, m):
        """c2w[:3, :3] @ unit_normals.reshape(3, -1)  (train.py:177-178; called by pose.SmallMatrix.__matmul__): deferred when the matrix
        IS the top-left block of the inverse pose the frame's camera remembers (pose.PoseMatrix) -- the fused loss forms its rotation
        from the same world_view_transform"""
        if self.kind == "unit_flat" and self.frame.view is not None:
            wvt = self.frame.view.world_view_transform
            cache = getattr(wvt, "__dict__", {}).get("_gof_inverse_of_T")
            if (cache is not None and cache[0] == wvt._version and tuple(m.shape) == (3, 3) and m.data_ptr() == cache[1].data_ptr()
                    and m.stride() == cache[1].stride() and m.dtype == torch.float32):
                self.frame.c2w33 = m
                return DeferredTensor(self.frame, "world_flat", self.shape, lambda: m @ self.eager())
        return m @ self.eager()


def _is_scalar_tensor(x):
    return type(x) in (torch.Tensor, torch.nn.Parameter) and x.dim() == 0 and x.is_floating_point()


class DeferredLoss(_Deferred):
    """sum_i coef[i] * term_i + const (+ extra) over the terms "l1", "ssim", "dn", "dist" of one frame (train.py:156-182).  `extra`: a 0-dim
    TENSOR the script added -- the decoupled-appearance L1 of train.py:158-159, which goes through a network and stays torch code -- carried
    along as it is (its own autograd graph): the deferred part is still one fused call, and the backward starts from both."""
    def __init__(self, frame, coef, const, extra=None):
        self.frame, self.coef, self.const, self.extra = frame, coef, const, extra

    # ---- python-number arithmetic: coefficients only (a carried tensor term: 0-dim torch arithmetic)
    def _scaled(self, k):
        return DeferredLoss(self.frame, {t: a * k for t, a in self.coef.items()}, self.const * k, None if self.extra is None else self.extra * k)

    def _plus(self, other, sign):
        if _is_number(other):
            return DeferredLoss(self.frame, dict(self.coef), self.const + sign * other, self.extra)
        if isinstance(other, DeferredLoss) and other.frame is self.frame:
            c = dict(self.coef)
            for t, a in other.coef.items():
                c[t] = c.get(t, 0.0) + sign * a
            extra = self.extra
            if other.extra is not None:
                extra = (other.extra if sign > 0 else -other.extra) if extra is None else (extra + other.extra if sign > 0 else extra - other.extra)
            return DeferredLoss(self.frame, c, self.const + sign * other.const, extra)
        if _is_scalar_tensor(other) and self.frame.image is not None and other.device == self.frame.image.device:
            term = other if sign > 0 else -other
            return DeferredLoss(self.frame, dict(self.coef), self.const, term if self.extra is None else self.extra + term)
        return NotImplemented

    def __add__(self, other):
        r = self._plus(other, 1.0)
        return self.tensor() + _materialize(other) if r is NotImplemented else r

    def __radd__(self, other):
        r = self._plus(other, 1.0)
        return _materialize(other) + self.tensor() if r is NotImplemented else r

    def __sub__(self, other):
        r = self._plus(other, -1.0)
        return self.tensor() - _materialize(other) if r is NotImplemented else r

    def __rsub__(self, other):
        if _is_number(other) or _is_scalar_tensor(other):
            r = self._scaled(-1.0)._plus(other, 1.0)
            if r is not NotImplemented:
                return r
        if isinstance(other, DeferredLoss):
            return other.__sub__(self)
        return _materialize(other) - self.tensor()

    def __mul__(self, other):
        return self._scaled(other) if _is_number(other) else self.tensor() * _materialize(other)

    def __rmul__(self, other):
        return self._scaled(other) if _is_number(other) else _materialize(other) * self.tensor()

    def __truediv__(self, other):
        return self._scaled(1.0 / other) if _is_number(other) and other != 0 else self.tensor() / _materialize(other)

    def __neg__(self):
        return self._scaled(-1.0)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        """`tensor + deferred`, `tensor - deferred`, `torch.add(...)`: the tensor's own operator reaches this hook -- a 0-dim tensor term
        is carried along (see `extra`); everything else computes the value"""
        if not kwargs and len(args) == 2:
            a, b = args
            if func in _ADDS:
                d, t = (a, b) if isinstance(a, DeferredLoss) else (b, a)
                if isinstance(d, DeferredLoss) and (_is_scalar_tensor(t) or _is_number(t)):
                    r = d._plus(t, 1.0)
                    if r is not NotImplemented:
                        return r
            elif func in _SUBS:
                if isinstance(a, DeferredLoss) and (_is_scalar_tensor(b) or _is_number(b)):
                    r = a._plus(b, -1.0)
                elif isinstance(b, DeferredLoss) and (_is_scalar_tensor(a) or _is_number(a)):
                    r = b._scaled(-1.0)._plus(a, 1.0)
                else:
                    r = NotImplemented
                if r is not NotImplemented:
                    return r
            elif func in _RSUBS and isinstance(b, DeferredLoss) and _is_scalar_tensor(a):       # Tensor.__rsub__(a, b) = b - a
                r = b._plus(a, -1.0)
                if r is not NotImplemented:
                    return r
        return func(*_materialize(args), **_materialize(kwargs or {}))

    # ---- evaluation
    def eager(self):
        """the combination as torch would have built it from the eager mirrors (differentiable)"""
        if self.frame.image is None:
            raise RuntimeError("this loss has been differentiated already (its frame's graph is gone); only its value is left")
        out = None
        for t, a in self.coef.items():
            v = self.frame.eager_term(t) * a
            out = v if out is None else out + v
        if self.extra is not None:
            out = self.extra if out is None else out + self.extra
        return out + self.const if self.const != 0 or out is None else out

    def tensor(self):
        """after the fused backward: the value (no graph); before: the eager combination"""
        f = self.frame
        if f.terms is None:
            return self.eager()
        out = torch.zeros((), dtype=torch.float32, device=f.terms["l1"].device) + self.const
        for t, a in self.coef.items():
            if a != 0:
                out = out + f.terms[t].detach() * a
        return out if self.extra is None else out + self.extra.detach()

    def item(self):
        f = self.frame
        if f.terms is None:
            return self.eager().item()
        if f.terms_host is None:
            names = list(f.terms)
            base = f.terms["l1"]._base
            if (base is not None and base.dim() == 1 and base.numel() == 6 and all(f.terms[n]._base is base for n in names)
                    and [f.terms[n].storage_offset() - base.storage_offset() for n in ("l1", "ssim", "dn", "dist")] == [1, 2, 4, 5]):
                v = base.tolist()                          # gof_train_loss's six terms are one tensor: ONE read-back, no launch
                f.terms_host = {"l1": v[1], "ssim": v[2], "dn": v[4], "dist": v[5]}
            else:
                f.terms_host = dict(zip(names, torch.stack([f.terms[n].detach() for n in names]).tolist()))
        return self.const + sum(a * f.terms_host[t] for t, a in self.coef.items() if a != 0) + (0.0 if self.extra is None else self.extra.item())

    def __float__(self):
        return float(self.item())

    def detach(self):
        return self.tensor().detach()

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        f = self.frame
        a = self.coef
        s = a.get("l1", 0.0) - a.get("ssim", 0.0)          # loss = s [(1 - l) L1 + l (1 - ssim) + l_dn dn + l_dist dist] + constant
        fused_ok = (gradient is None and not retain_graph and not create_graph and inputs is None and f.terms is None and not f._eager
                    and s > 0 and f.gt is not None and f.view is not None and (a.get("dn", 0.0) == 0 or f.dn_err is not None))
        if not fused_ok:
            return self.eager().backward(gradient=gradient, retain_graph=retain_graph, create_graph=create_graph, inputs=inputs)
        if abs(s - 1.0) < 1e-12:
            s = 1.0
        image = _plain(f.image)
        values, dL = impl["fused"](image, f.gt, f.view, -a.get("ssim", 0.0) / s, a.get("dn", 0.0) / s, a.get("dist", 0.0) / s)
        if s != 1.0:
            dL.mul_(s)
        f.terms = {"l1": values.Ll1, "ssim": values.ssim, "dn": values.depth_normal_loss, "dist": values.distortion_loss}
        # the frame's graph is not needed again: drop what ties the image, its slices and the deferred intermediates into reference
        # cycles (they would keep the forward's buffers alive until the cycle collector runs)
        f.image.__dict__.pop("_gof_frame", None)
        f._eager, f.dn_err, f.c2w33, f.image = {"consumed": True}, None, None, None
        stats["fused_backwards"] += 1
        if self.extra is not None and self.extra.requires_grad:
            torch.autograd.backward([image, self.extra], [dL, torch.ones_like(self.extra)])      # one pass over the graph: the rasterizer's backward runs once
        else:
            torch.autograd.backward(image, dL)


# ---- what the image's channel slices become -------------------------------------------------------------------------------------
_ADDS = (torch.add, torch.Tensor.add, torch.Tensor.__add__, torch.Tensor.__radd__)
_SUBS = (torch.sub, torch.Tensor.sub, torch.Tensor.__sub__)
_RSUBS = (torch.Tensor.__rsub__,)


class RenderedChannels(torch.Tensor):
    """Whole channels [lo, hi) of a RenderedImage (diff_gaussian_rasterization): a plain tensor that remembers which, so that the
    three uses train.py makes of them -- rendering[8].mean(), depth[None, ...] on its way to depth_to_normal, F.normalize of
    rendering[3:6] -- can answer with a deferred value.  Every other operation gives plain tensors."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if _ENABLED and torch.is_grad_enabled() and args and isinstance(args[0], RenderedChannels):
            r = _match_channels(func, args, kwargs)
            if r is not NotImplemented:
                return r
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        return out.as_subclass(torch.Tensor) if isinstance(out, RenderedChannels) else out


def _prov(t):
    return t.__dict__.get("_gof_prov") if isinstance(t, RenderedChannels) else None


def _match_channels(func, args, kwargs):
    x = args[0]
    p = _prov(x)
    if p is None:
        return NotImplemented
    frame, lo, hi, squeeze, lifted = p
    if func is torch.Tensor.mean and len(args) == 1 and not kwargs and (lo, hi, squeeze, lifted) == (8, 9, True, False):
        return DeferredLoss(frame, {"dist": 1.0}, 0.0)                                        # train.py:164-167
    if func is torch.Tensor.__getitem__ and len(args) == 2 and (lo, hi, squeeze, lifted) == (6, 7, True, False):
        idx = args[1]
        if idx is None or (isinstance(idx, tuple) and len(idx) == 2 and idx[0] is None and idx[1] is Ellipsis):      # depth[None, ...]: train.py:171
            with torch._C.DisableTorchFunctionSubclass():
                out = torch.Tensor.__getitem__(x, idx).as_subclass(RenderedChannels)
            out._gof_prov = (frame, lo, hi, squeeze, True)
            return out
    if func is torch.nn.functional.normalize and len(args) == 1 and (lo, hi, squeeze, lifted) == (3, 6, False, False):
        if kwargs.get("p", 2.0) == 2 and kwargs.get("dim", 1) == 0 and kwargs.get("eps", 1e-12) == 1e-12 and kwargs.get("out") is None:      # train.py:175
            return DeferredTensor(frame, "unit", x.shape, lambda: torch.nn.functional.normalize(_plain(x), p=2, dim=0))
    return NotImplemented


def _on_slice(out, image, lo, hi, squeeze):
    """diff_gaussian_rasterization.RenderedImage: called with every whole-channel slice it hands out"""
    if not _ENABLED:
        return out
    with torch._C.DisableTorchFunctionSubclass():
        out = out.as_subclass(RenderedChannels)
    out._gof_prov = (_Frame.of(image), lo, hi, squeeze, False)
    return out


# ---- the three helpers the launcher rebinds ---------------------------------------------------------------------------------------
def l1_loss(network_output, gt):
    """utils/loss_utils.py:17-18; deferred for (rendering[:3], ground truth) -- train.py:156"""
    p = _prov(network_output) if _ENABLED and torch.is_grad_enabled() else None
    if p is not None and p[1:] == (0, 3, False, False) and p[0].accept_gt(network_output, gt):
        return DeferredLoss(p[0], {"l1": 1.0}, 0.0)
    return impl["l1"](network_output, gt)


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:30-41; deferred for (rendering[:3], the same ground truth) -- train.py:161"""
    p = _prov(img1) if _ENABLED and torch.is_grad_enabled() else None
    if p is not None and p[1:] == (0, 3, False, False) and window_size == 11 and size_average is True and p[0].accept_gt(img1, img2):
        return DeferredLoss(p[0], {"ssim": 1.0}, 0.0)
    return impl["ssim"](img1, img2, window_size, size_average)


def depth_to_normal(view, depth):
    """utils/depth_utils.py:24-35; deferred for (the frame's camera, rendering[6][None, ...]) -- train.py:170-172"""
    p = _prov(depth) if _ENABLED and torch.is_grad_enabled() else None
    if p is not None and p[1:] == (6, 7, True, True):
        frame = p[0]
        H, W = int(frame.image.shape[1]), int(frame.image.shape[2])
        wvt = getattr(view, "world_view_transform", None)
        if (frame.view is None and isinstance(wvt, torch.Tensor) and not wvt.requires_grad and int(getattr(view, "image_width", -1)) == W
                and int(getattr(view, "image_height", -1)) == H):
            frame.view = view
            pair = []

            def both():
                if not pair:
                    pair.extend(impl["depth_to_normal"](view, _plain(depth)))
                return pair
            return (DeferredTensor(frame, "dn_hw3", (H, W, 3), lambda: both()[0]), DeferredTensor(frame, "points_hw3", (H, W, 3), lambda: both()[1]))
    return impl["depth_to_normal"](view, depth)


def enable(on=True):
    """Deferred evaluation for the images rendered from now on (the launcher calls this before the script starts)."""
    global _ENABLED
    import diff_gaussian_rasterization as DGR
    _ENABLED = bool(on)
    DGR._slice_hook = _on_slice if _ENABLED else None
