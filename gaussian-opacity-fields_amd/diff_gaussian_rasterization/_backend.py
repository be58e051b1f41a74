"""ctypes backend: the four entry points of the reference's pybind11 module
``diff_gaussian_rasterization._C`` (reference submodules/diff-gaussian-rasterization/ext.cpp:15-20,
rasterize_points.cu) implemented on top of the C ABI of ``libgof_hip.so`` (include/gof_hip.h).

PyTorch is used for device memory, streams and autograd plumbing only; every kernel lives in
the shared library.  There is NO fallback: if the library is missing or cannot be loaded,
importing this module raises.
"""
import ctypes as C
import os
import weakref
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("GOF_HIP_LIB", os.path.join(_PKG, "lib", "libgof_hip.so"))

OUTPUT_CHANNELS = 9


class GofRasterArgs(C.Structure):
    """Mirror of ``GofRasterArgs`` in include/gof_hip.h."""
    _fields_ = [
        ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("kernel_size", C.c_float), ("scale_modifier", C.c_float),
        ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("background", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
        ("view2gaussian_precomp", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("campos", C.c_void_p), ("subpixel_offset", C.c_void_p), ("shs_rest", C.c_void_p),
        ("forward_exact", C.c_int32), ("tight_tile_rects", C.c_int32), ("integrate_pixel_pass", C.c_int32), ("reserved0", C.c_int32),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libgof_hip.so not found at %s -- build it with `python gaussian-opacity-fields_amd/build.py` "
            "(hipcc, gfx950). There is no CPU / PyTorch fallback for the rasterizer." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, sz, i32, u32, i64 = C.c_void_p, C.c_size_t, C.c_int32, C.c_uint32, C.c_int64
    A = C.POINTER(GofRasterArgs)
    lib.gof_last_error.restype = C.c_char_p
    lib.gof_abi_version.restype = C.c_int
    for name, args in (("gof_geom_bytes", [i32]), ("gof_geom_bytes_forward", [i32]), ("gof_image_bytes", [i32, i32]), ("gof_binning_bytes", [u32, i32, i32]), ("gof_binning_bytes_for", [u32, i32, i32, u32]),
                       ("gof_point_bytes", [i32]), ("gof_backward_scratch_bytes", [i32, u32]), ("gof_backward_scratch_bytes_for", [i32, u32, u32]),
                       ("gof_mtets_tet_ws_bytes", [i64]),
                       ("gof_mtets_edge_ws_bytes", [i64])):
        f = getattr(lib, name)
        f.restype = sz
        f.argtypes = args
    lib.gof_forward_prepare.argtypes = lib.gof_integrate_prepare.argtypes = [A, vp, sz, vp, sz, vp, C.POINTER(u32), vp]
    lib.gof_forward_render.argtypes = [A, u32, vp, vp, sz, vp, sz, vp, sz, vp, vp]
    lib.gof_forward_fused.argtypes = [A, u32, vp, sz, vp, sz, vp, sz, vp, vp, vp, vp, vp]
    lib.gof_forward_fused.restype = C.c_int
    lib.gof_backward.argtypes = [A, u32, vp, vp, sz, vp, sz, vp, sz, vp] + [vp] * 10 + [vp, sz, vp]
    lib.gof_backward_query.argtypes = [A, u32, sz, vp, sz, C.POINTER(u32), vp]
    lib.gof_forward_usage_async.argtypes = [A, vp, sz, vp, vp]
    lib.gof_usage_decode.argtypes = [vp, u32, i32, i32, sz, C.POINTER(u32)]
    lib.gof_backward_query.restype = lib.gof_forward_usage_async.restype = lib.gof_usage_decode.restype = C.c_int
    lib.gof_backward_blend.argtypes = lib.gof_backward_preprocess.argtypes = lib.gof_backward.argtypes
    lib.gof_backward_blend.restype = lib.gof_backward_preprocess.restype = C.c_int
    lib.gof_integrate_prepare_points.argtypes = [A, i32, vp, vp, sz, C.POINTER(u32), vp]
    lib.gof_integrate_run.argtypes = [A, u32, vp, i32, u32, vp, sz, vp, sz, vp, sz, vp, sz, vp, sz, vp, vp, vp, vp]
    lib.gof_integrate_view.argtypes = [A, u32, vp, vp, sz, vp, sz, vp, sz, vp, vp]
    lib.gof_integrate_points.argtypes = [A, u32, i32, u32, vp, sz, vp, sz, vp, sz, vp, sz, vp, sz, vp, vp, vp, vp, vp]
    lib.gof_integrate_points_packed.argtypes = lib.gof_integrate_points.argtypes
    lib.gof_integrate_points_min.argtypes = [A, u32, i32, u32, i32, vp, sz, vp, sz, vp, sz, vp, sz, vp, sz, vp, vp, vp, vp, vp]
    lib.gof_integrate_pack_geom.argtypes = [A, vp, sz, vp, sz, vp]
    lib.gof_integrate_packed_geom_bytes.restype = sz
    lib.gof_integrate_packed_geom_bytes.argtypes = [i32]
    lib.gof_point_binning_bytes.restype = sz
    lib.gof_point_binning_bytes.argtypes = [u32, i32, i32]
    lib.gof_mark_visible.argtypes = [i32, vp, vp, vp, vp, vp]
    lib.gof_sh_grad_pack.argtypes = [i32, vp, vp, sz, vp, vp, vp]
    lib.gof_sh_grad_expand.argtypes = [i32, i32, i32, i32, vp, vp, i64, vp, i64, C.c_float, vp, i64, vp, i64, vp]
    lib.gof_sh_grad_pack.restype = lib.gof_sh_grad_expand.restype = C.c_int
    lib.gof_mtets_classify.argtypes = [i64, i64, vp, vp, vp, sz, C.POINTER(i64), vp]
    lib.gof_mtets_count.argtypes = [i64, i64, vp, vp, vp, sz, vp, sz, C.POINTER(i64), C.POINTER(i64), vp]
    lib.gof_mtets_emit.argtypes = [i64, i64, vp, vp, vp, vp, vp, sz, vp, sz, i64, i64, vp, vp, vp, vp, vp, vp]
    lib.gof_debug_fetch.restype = i64
    lib.gof_debug_fetch.argtypes = [C.c_char_p, A, u32, vp, vp, vp, vp, sz, vp]
    lib.gof_set_forward_exact.argtypes = lib.gof_set_tight_tile_rects.argtypes = lib.gof_set_integrate_pixel_pass.argtypes = [C.c_int]
    lib.gof_set_forward_exact.restype = lib.gof_set_tight_tile_rects.restype = lib.gof_set_integrate_pixel_pass.restype = C.c_int
    lib.gof_profile_enable.argtypes = [C.c_int]
    lib.gof_profile_report.argtypes = [C.c_char_p, sz]
    for name in ("gof_profile_enable", "gof_profile_report", "gof_forward_prepare", "gof_integrate_prepare", "gof_forward_render", "gof_backward", "gof_integrate_prepare_points",
                 "gof_integrate_run", "gof_integrate_view", "gof_integrate_points", "gof_integrate_points_packed", "gof_integrate_points_min", "gof_integrate_pack_geom", "gof_mark_visible", "gof_mtets_classify", "gof_mtets_count", "gof_mtets_emit"):
        getattr(lib, name).restype = C.c_int
    return lib


lib = _load()


_grad_bucket = (None, 0, 0)   # (weak reference to the storage, first byte, one past the last byte) of the latest backward's gradient allocation


def is_in_grad_bucket(t):
    """True if `t` lies in the gradient allocation of the most recent rasterizer backward (the segments of its parameter gradients).
    train_epilogue/activations.py writes the raw-parameter gradient over an incoming gradient only then: such a tensor is this
    library's own scratch, not a gradient autograd shares between nodes."""
    # identity AND liveness of the storage, not its address: once the bucket is freed the caching allocator may hand the same block to
    # an unrelated gradient (a weak reference to a storage dies with the last tensor that uses it; torch keeps one Python object per
    # live storage, so `is` compares storages)
    ref, lo, hi = _grad_bucket
    st = ref() if ref is not None else None
    return st is not None and t.untyped_storage() is st and lo <= t.data_ptr() and t.data_ptr() + t.numel() * t.element_size() <= hi


def _check(rc):
    if rc != 0:
        raise RuntimeError("libgof_hip: " + lib.gof_last_error().decode(errors="replace"))


def _ptr(t):
    """Device pointer of a tensor, or NULL for the reference's "absent" convention (an empty tensor)."""
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())


def _dev_f32(t, device, what):
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        raise RuntimeError("%s must be on %s (got %s)" % (what, device, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (what, t.dtype))
    return t.contiguous()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # the current stream's handle without building a torch.cuda.Stream object


def _stream():
    """The caller's stream: every launch of a call goes to the CURRENT stream of the current device (ten lookups per training iteration:
    torch.cuda.current_stream() costs 4-14 us each, the raw query a fraction of that)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_call_modes = threading.local()


class call_modes:
    """``with call_modes(forward_exact=True): ...`` -- the modes of the rasterizer calls made by THIS thread inside the block, handed
    to the library per call (GofRasterArgs.forward_exact / tight_tile_rects / integrate_pixel_pass, ABI 12): None = the process-wide
    default (set_forward_exact & co. below), True / False = on / off for these calls whatever the default.  Nothing process-wide
    changes: other threads, and other streams of this thread, keep their own modes.  Blocks nest; an inner None keeps the outer value."""

    def __init__(self, forward_exact=None, tight_tile_rects=None, integrate_pixel_pass=None):
        self.want = {"forward_exact": forward_exact, "tight_tile_rects": tight_tile_rects, "integrate_pixel_pass": integrate_pixel_pass}

    def __enter__(self):
        self.prev = getattr(_call_modes, "value", None)
        cur = dict(self.prev or {})
        cur.update({k: v for k, v in self.want.items() if v is not None})
        _call_modes.value = cur
        return self

    def __exit__(self, *exc):
        _call_modes.value = self.prev
        return False


def _mode_field(name):
    v = (getattr(_call_modes, "value", None) or {}).get(name)
    return 0 if v is None else (1 if v else -1)


class _View:
    """Validated, contiguous inputs of one rasterization call + the POD handed to the library."""

    def __init__(self, background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                 view2gaussian_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, subpixel_offset,
                 image_height, image_width, sh, degree, campos, prefiltered, debug):
        if means3D.dim() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")   # rasterize_points.cu:61-63
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError("diff_gaussian_rasterization (gfx950 backend) needs tensors on a ROCm device, got %s" % dev)
        self.device = dev
        self.P = int(means3D.size(0))
        self.H, self.W = int(image_height), int(image_width)
        k = self.keep = {}
        k["bg"] = _dev_f32(background, dev, "bg")
        k["means3D"] = _dev_f32(means3D, dev, "means3D")
        k["colors"] = _dev_f32(colors, dev, "colors_precomp")
        k["opacity"] = _dev_f32(opacity, dev, "opacities")
        k["scales"] = _dev_f32(scales, dev, "scales")
        k["rotations"] = _dev_f32(rotations, dev, "rotations")
        k["cov3D"] = _dev_f32(cov3D_precomp, dev, "cov3D_precomp")
        k["v2g"] = _dev_f32(view2gaussian_precomp, dev, "view2gaussian_precomp")
        k["view"] = _dev_f32(viewmatrix, dev, "viewmatrix")
        k["proj"] = _dev_f32(projmatrix, dev, "projmatrix")
        k["campos"] = _dev_f32(campos, dev, "campos")
        k["subpix"] = _dev_f32(subpixel_offset, dev, "subpixel_offset")
        # sh: [P,M,3] as the reference passes it, or the pair (features_dc [P,1,3], features_rest [P,15,3]) as the reference STORES
        # it (scene/gaussian_model.py:351-352; GofRasterArgs.shs_rest) -- no 192 B/Gaussian concatenation per iteration
        self.split_sh = isinstance(sh, (tuple, list))
        if self.split_sh:
            dc, rest = sh
            if dc.dim() != 3 or rest.dim() != 3 or tuple(dc.shape) != (self.P, 1, 3) or tuple(rest.shape) != (self.P, 15, 3):
                raise RuntimeError("separate SH tensors must be (P,1,3) and (P,15,3), got %s and %s" % (tuple(dc.shape), tuple(rest.shape)))
            k["sh"] = _dev_f32(dc, dev, "sh (DC)")
            k["sh_rest"] = _dev_f32(rest, dev, "sh (higher bands)")
            self.M = 16
        else:
            k["sh"] = _dev_f32(sh, dev, "sh")
            k["sh_rest"] = None
            self.M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0   # rasterize_points.cu:87-91
        a = self.args = GofRasterArgs()
        a.P, a.D, a.M, a.W, a.H = self.P, int(degree), self.M, self.W, self.H
        a.tan_fovx, a.tan_fovy = float(tan_fovx), float(tan_fovy)
        a.kernel_size, a.scale_modifier = float(kernel_size), float(scale_modifier)
        a.prefiltered, a.debug = int(bool(prefiltered)), int(bool(debug))
        a.background = _ptr(k["bg"]); a.means3D = _ptr(k["means3D"]); a.shs = _ptr(k["sh"])
        a.colors_precomp = _ptr(k["colors"]); a.opacities = _ptr(k["opacity"])
        a.scales = _ptr(k["scales"]); a.rotations = _ptr(k["rotations"])
        a.cov3D_precomp = _ptr(k["cov3D"]); a.view2gaussian_precomp = _ptr(k["v2g"])
        a.viewmatrix = _ptr(k["view"]); a.projmatrix = _ptr(k["proj"]); a.campos = _ptr(k["campos"])
        a.subpixel_offset = _ptr(k["subpix"])
        a.shs_rest = _ptr(k["sh_rest"]) if k["sh_rest"] is not None else None
        a.forward_exact, a.tight_tile_rects, a.integrate_pixel_pass = _mode_field("forward_exact"), _mode_field("tight_tile_rects"), _mode_field("integrate_pixel_pass")

    def ref(self):
        return C.byref(self.args)

    def bytes_tensor(self, n):
        return torch.empty(int(n), dtype=torch.uint8, device=self.device)


def _prepare_and_bin(v, for_query=False):
    """Stage 1 shared by forward and integrate: preprocess + scan + instance count.  for_query: the footprints complete (pixel box,
    front depth: gof_integrate_prepare), which only the opacity-field query reads."""
    geom = v.bytes_tensor(lib.gof_geom_bytes(v.P) if for_query else lib.gof_geom_bytes_forward(v.P))      # (the query's part of the footprints: 16 B per Gaussian more)
    img = v.bytes_tensor(lib.gof_image_bytes(v.W, v.H))
    radii = torch.empty(v.P, dtype=torch.int32, device=v.device)       # preprocess_fwd writes every element (0 for culled Gaussians)
    n = C.c_uint32(0)
    _check((lib.gof_integrate_prepare if for_query else lib.gof_forward_prepare)(v.ref(), _ptr(geom), geom.numel(), _ptr(img), img.numel(), _ptr(radii), C.byref(n), _stream()))
    rendered = int(n.value)
    binning = v.bytes_tensor(lib.gof_binning_bytes(rendered, v.W, v.H))
    return geom, img, binning, radii, rendered


GOF_E_CAPACITY = -5
FUSED_FORWARD = os.environ.get("GOF_FUSED_FORWARD", "1") != "0"
FULL_BACKWARD_SCRATCH = os.environ.get("GOF_FULL_BACKWARD_SCRATCH", "0") == "1"
FULL_MASK_POOL = os.environ.get("GOF_FULL_MASK_POOL", "0") == "1"
_capacity = {}          # (device, P, W, H) -> instance capacity learnt from earlier frames
_mask_need = {}         # (device, P, W, H) -> most contributor-mask sub-chunks a forward of this shape has asked for (learnt at its backward)
_staged_need = {}       # (device, P, W, H) -> most tile-list entries a backward of this shape has staged (= partial gradient records written)
USAGE_WORDS = 66        # GOF_USAGE_WORDS (include/gof_hip.h)
_stats = {"fused_redone_frames": 0, "last_num_rendered": 0, "mask_pool_redone_frames": 0, "record_pool_redone_backwards": 0,
          "backward_queries": 0, "two_stage_frames": 0, "inherited_shapes": 0, "forwards": 0, "backwards": 0}      # bench.py reads these (no effect on the path)
_recent_P = {}          # (device, W, H) -> P of the latest frame at that resolution


def _inherit_learnt(shape_key):
    """Training changes P at every densification (train.py:258-264, every 100 iterations): a shape never seen before would run the
    two-stage forward (a host read-back mid-frame) and a synchronising backward query, and would learn its pools from scratch over the
    next views.  A new P at a resolution whose previous frames had between half and twice as many Gaussians INHERITS what those
    frames learnt -- instance capacity, mask sub-chunks, staged records -- scaled by the growth (never down: a capacity above the
    count costs nothing); the pools are verified per frame as ever, so a guess that turns out too small costs one
    redone frame.  The old shape's entries are dropped (P changes for good in training; the dicts stay bounded)."""
    dev, P, W, H = shape_key
    prev = _recent_P.get((dev, W, H))
    _recent_P[(dev, W, H)] = P
    if prev is None or prev == P or shape_key in _capacity:
        return
    old = (dev, prev, W, H)
    if old not in _capacity or not (0.5 * prev <= P <= 2.0 * prev):
        return
    grow = max(1.0, P / float(prev))
    _capacity[shape_key] = (int(_capacity.pop(old) * grow) + 0xFFFF) & ~0xFFFF
    for d in (_mask_need, _staged_need):
        if old in d:
            d[shape_key] = int(d.pop(old) * grow) + 1
    _stats["inherited_shapes"] += 1


class MaskPoolTooSmall(RuntimeError):
    """Raised by rasterize_gaussians_backward BEFORE anything is launched: the forward of this frame asked for more contributor-mask
    sub-chunks than its binning workspace held (include/gof_hip.h: gof_binning_bytes_for), so masks are missing.  The autograd
    functions repeat the frame's forward with a full pool and call the backward again; the learnt need has been raised."""

    def __init__(self, requested, capacity):
        super().__init__("contributor-mask pool too small: %d sub-chunks requested, %d held" % (requested, capacity))
        self.requested, self.capacity = requested, capacity


def _exchange_starts_inside_backward():
    """A data-parallel reducer starts its all-gather from inside the backward (set_sh_grad_ready_callback), right after the blend
    stage.  Until round 4 such frames ran on WORST-CASE pools (118 B per instance): a backward repeated after the colour gradient
    has gone on the wire would be wrong, and a read-back in front of the backward idles every rank's GPU.  Round 5: the same
    optimistic pools as on one GPU -- the blend stage is queued, the host then looks at the FORWARD's counters (on their way to pinned
    memory since the forward's last kernel; the GPU is busy with the blend meanwhile), repeats the blend stage if its record pool was
    too small (or raises MaskPoolTooSmall: forward again), and only then hands the colour gradient to the reducer
    (rasterize_gaussians_backward)."""
    return _sh_track["on"] and _sh_track["ready_cb"] is not None


def _mask_pool_subchunks(shape_key):
    """sub-chunks to size the fused forward's mask pool for: 1.25 x the largest request seen for this shape (None: not learnt yet,
    or switched off -> the worst case)"""
    need = None if FULL_MASK_POOL else _mask_need.get(shape_key)
    return None if need is None else int(need * 1.25) + 256
_pinned = threading.local()      # .by_device: device -> pinned host word for the asynchronous instance-count read-back (per thread)


def _round_capacity(n):
    return (int(n * 1.25) + (1 << 16)) & ~0xFFFF


class NumRendered(int):
    """``num_rendered`` as the reference returns it -- the instance count of the frame (rasterize_points.cu:119) -- that also
    remembers the CAPACITY the frame's binning workspace was laid out for when the sync-free forward ran (``layout``; equal to
    the count on the two-stage path).  The backward needs the layout size to find its arrays; everything else sees the count.
    ``usage``: (pinned words, event) of the frame's pool counters on their way to the host (gof_forward_usage_async), or None."""

    def __new__(cls, count, layout=None, usage=None):
        self = super().__new__(cls, count)
        self.layout = int(count if layout is None else layout)
        self.usage = usage
        return self

    def __del__(self):                      # the frame is gone (its autograd graph was freed): its pinned words can serve another frame
        u = getattr(self, "usage", None)
        if u is not None and len(_usage_free) < 64:
            _usage_free.append(u[0])


_usage_free = []        # pinned host buffers (USAGE_WORDS int32 each) not attached to a live frame


_zeros = {}


def _zero_scalar(dev):
    """one zero per device, the storage of every stride-0 dL_dcov3D (a fill kernel per backward otherwise)"""
    z = _zeros.get(dev)
    if z is None:
        z = _zeros[dev] = torch.zeros(1, dtype=torch.float32, device=dev)
    return z


def _layout_count(num_rendered):
    return int(getattr(num_rendered, "layout", num_rendered))


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        view2gaussian_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, subpixel_offset,
                        image_height, image_width, sh, degree, campos, prefiltered, debug, fused=None):
    """Replaces ``_C.rasterize_gaussians`` (RasterizeGaussiansCUDA, rasterize_points.cu:36-122).
    Returns ``(num_rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer)``.

    The first frame of a (device, P, W, H) shape runs the reference's two-stage forward (instance count read back in the middle to
    size the binning buffer).  Later frames size that buffer for 1.25 x the last count and run ``gof_forward_fused`` -- no pipeline
    bubble; the returned ``num_rendered`` is the frame's true instance count and carries, as ``.layout``, the capacity that fixes
    the workspace layout for the backward (NumRendered); a frame that needs more is redone through the two-stage path, transparently.  ``fused=False`` / ``GOF_FUSED_FORWARD=0`` disable this."""
    v = _View(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, view2gaussian_precomp,
              viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, subpixel_offset, image_height, image_width, sh,
              degree, campos, prefiltered, debug)
    use_fused = FUSED_FORWARD if fused is None else bool(fused)
    with torch.cuda.device(v.device):
        out_color = torch.empty((OUTPUT_CHANNELS, v.H, v.W), dtype=torch.float32, device=v.device)
        if v.P == 0:
            out_color.zero_()
            empty = v.bytes_tensor(0)
            return 0, out_color, torch.zeros(0, dtype=torch.int32, device=v.device), empty, empty.clone(), empty.clone()
        shape_key = (str(v.device), v.P, v.W, v.H)
        _stats["forwards"] += 1
        if use_fused and not prefiltered and not debug:
            _inherit_learnt(shape_key)
        cap = _capacity.get(shape_key) if (use_fused and not prefiltered and not debug) else None
        if cap is not None:
            geom = v.bytes_tensor(lib.gof_geom_bytes_forward(v.P))
            img = v.bytes_tensor(lib.gof_image_bytes(v.W, v.H))
            sub = _mask_pool_subchunks(shape_key)
            binning = v.bytes_tensor(lib.gof_binning_bytes(cap, v.W, v.H) if sub is None else lib.gof_binning_bytes_for(cap, v.W, v.H, sub))
            radii = torch.empty(v.P, dtype=torch.int32, device=v.device)
            # one pinned count word per (thread, device): ctypes releases the GIL for the call, and the library writes / waits for / reads
            # this word inside it -- two threads rendering on the same device must not share it
            pins = getattr(_pinned, "by_device", None)
            if pins is None:
                pins = _pinned.by_device = {}
            pin = pins.get(str(v.device))
            if pin is None:
                pin = pins[str(v.device)] = torch.zeros(4, dtype=torch.int32).pin_memory()
            # the frame's pool counters go to pinned host memory at the end of the forward (stored by its last kernel: no copy launch),
            # an event behind the call tells the backward when they are there (optimistic pools, rasterize_gaussians_backward)
            no_counters = FULL_MASK_POOL and FULL_BACKWARD_SCRATCH
            words = None if no_counters else (_usage_free.pop() if _usage_free else torch.empty(USAGE_WORDS, dtype=torch.int32).pin_memory())      # (pinning costs tens of microseconds: recycled)
            rc = lib.gof_forward_fused(v.ref(), cap, _ptr(geom), geom.numel(), _ptr(binning), binning.numel(), _ptr(img), img.numel(),
                                       _ptr(radii), _ptr(out_color), C.c_void_p(pin.data_ptr()), None if words is None else C.c_void_p(words.data_ptr()), _stream())
            usage = None
            if words is not None:
                ev = torch.cuda.Event()
                ev.record()
                usage = (words, ev)
            if rc == 0:
                true_r = int(pin[0].item()) & 0xFFFFFFFF
                if _round_capacity(true_r) > cap:
                    _capacity[shape_key] = _round_capacity(true_r)        # growing scene: stay ahead of it
                _stats["last_num_rendered"] = true_r
                return NumRendered(true_r, cap, usage), out_color, radii, geom, binning, img
            if rc != GOF_E_CAPACITY:
                _check(rc)
            _stats["fused_redone_frames"] += 1
            if words is not None:
                _usage_free.append(words)
            del geom, img, binning, radii, usage                                     # too small: redo the frame below with the exact count
        geom, img, binning, radii, rendered = _prepare_and_bin(v)
        _stats["two_stage_frames"] += 1               # (a host read-back in the middle of the frame, as the reference: rasterizer_impl.cu:336)
        _check(lib.gof_forward_render(v.ref(), rendered, _ptr(radii), _ptr(geom), geom.numel(), _ptr(binning), binning.numel(),
                                      _ptr(img), img.numel(), _ptr(out_color), _stream()))
        if use_fused and not prefiltered and not debug:
            _capacity[shape_key] = max(_capacity.get(shape_key, 0), _round_capacity(rendered))
        _stats["last_num_rendered"] = int(rendered)
    return NumRendered(rendered), out_color, radii, geom, binning, img


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 view2gaussian_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size,
                                 subpixel_offset, dL_dout_color, sh, degree, campos, geomBuffer, R, binningBuffer,
                                 imageBuffer, debug):
    """Replaces ``_C.rasterize_gaussians_backward`` (rasterize_points.cu:124-211).  Returns
    ``(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations,
    dL_dview2gaussian)``."""
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    # opacity is not an input of the reference's backward; the library needs a non-NULL pointer only for validation
    v = _View(background, means3D, colors, means3D, scales, rotations, scale_modifier, cov3D_precomp, view2gaussian_precomp,
              viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, subpixel_offset, H, W, sh, degree, campos, False, debug)
    P, M, dev = v.P, v.M, v.device
    f = dict(dtype=torch.float32, device=dev)
    # The gradients of the Gaussian PARAMETERS (means3D, opacity, scales, rotations, sh: 59 floats per Gaussian at SH degree 3)
    # are carved from ONE allocation in that order, 16-byte aligned segments: a data-parallel trainer can all-reduce the
    # bucket in place (dp/reducer.py) instead of packing 236 B per Gaussian into a bucket and back -- or only its first four
    # segments when the SH gradient travels in compressed form (sh_grad_pack / sh_grad_expand below).
    sizes = [3 * P, P, 3 * P, 4 * P] + ([3 * P, 45 * P] if v.split_sh else [3 * M * P])
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += (n + 3) & ~3
    bucket = torch.empty(tot, **f)
    global _grad_bucket
    _grad_bucket = (weakref.ref(bucket.untyped_storage()), bucket.data_ptr(), bucket.data_ptr() + 4 * tot)     # what is_in_grad_bucket() recognises
    g_means3D = bucket[offs[0]:offs[0] + sizes[0]].view(P, 3); g_opacity = bucket[offs[1]:offs[1] + sizes[1]].view(P, 1)
    g_scales = bucket[offs[2]:offs[2] + sizes[2]].view(P, 3); g_rot = bucket[offs[3]:offs[3] + sizes[3]].view(P, 4)
    if v.split_sh:        # gradients in the layout of the inputs: (dL_dfeatures_dc [P,1,3], dL_dfeatures_rest [P,15,3])
        g_sh = (bucket[offs[4]:offs[4] + sizes[4]].view(P, 1, 3), bucket[offs[5]:offs[5] + sizes[5]].view(P, 15, 3))
    else:
        g_sh = bucket[offs[4]:offs[4] + sizes[4]].view(P, M, 3)
    # dL_dview2gaussian / dL_dmeans2D / dL_dcolors are written completely by the library (gather_tile_partials).  dL_dcov3D is a DEAD
    # output of the reference (always zero: its consumer is commented out, backward.cu:627-630, rasterize_points.cu:166): without a
    # cov3D_precomp to hand it to, it is returned as a stride-0 view of one zero -- same values, no 24 B per Gaussian to allocate and
    # clear every iteration; with cov3D_precomp a real zero tensor is produced (the library clears it).
    acc = torch.empty(16 * P, **f)
    g_v2g = acc[:10 * P].view(P, 10)
    g_means2D = acc[10 * P:13 * P].view(P, 3); g_colors = acc[13 * P:16 * P].view(P, 3)
    want_cov3D = isinstance(cov3D_precomp, torch.Tensor) and cov3D_precomp.numel() > 0
    g_cov3D = torch.empty((P, 6), **f) if want_cov3D else _zero_scalar(dev).expand(P, 6)
    if P != 0:
        dl = _dev_f32(dL_dout_color, dev, "dL_dout_color")
        _stats["backwards"] += 1
        with torch.cuda.device(dev):
            # Pools (round 4).  The record pool of the scratch needs as many records as the forward staged entries (~30 % of R at S1M);
            # the frame's mask pool (in binningBuffer) was sized from earlier frames.  Both numbers are on their way to pinned host
            # memory since the end of the forward (NumRendered.usage).  OPTIMISTIC: size the record pool 1.25 x the most any earlier
            # backward of this shape staged, launch, and only then look at the counters -- the host waits while the GPU already works
            # on the backward; a pool that was too small (a new view that needs more than anything seen so far) drops what does not
            # fit, and the backward is repeated with the exact size (or, for missing masks, after the frame's forward was repeated:
            # MaskPoolTooSmall).  Without counters in flight (first frame of a shape, two-stage forward): the synchronising query.
            shape_key = (str(dev), P, W, H)
            usage = getattr(R, "usage", None)
            R = _layout_count(R)
            full_pool = binningBuffer.numel() >= lib.gof_binning_bytes(int(R), W, H)
            staged_guess = _staged_need.get(shape_key)
            verify = None
            if FULL_BACKWARD_SCRATCH and full_pool:
                nscratch = lib.gof_backward_scratch_bytes(P, int(R))          # a record per instance, a full mask pool: nothing can be missing
            elif usage is not None and staged_guess is not None and not FULL_BACKWARD_SCRATCH:
                rec_guess = min(int(R), int(staged_guess * 1.25) + 4096)
                nscratch = lib.gof_backward_scratch_bytes_for(P, int(R), rec_guess)
                verify = (usage, rec_guess)
            else:
                q = (C.c_uint32 * 3)()
                _stats["backward_queries"] += 1           # (a synchronising read-back: first frame of a shape, two-stage forward)
                _check(lib.gof_backward_query(v.ref(), int(R), binningBuffer.numel(), _ptr(imageBuffer), imageBuffer.numel(), q, _stream()))
                staged, requested, held = int(q[0]), int(q[1]), int(q[2])
                _mask_need[shape_key] = max(_mask_need.get(shape_key, 0), requested)
                _staged_need[shape_key] = max(_staged_need.get(shape_key, 0), staged)
                if requested > held:
                    raise MaskPoolTooSmall(requested, held)
                nscratch = lib.gof_backward_scratch_bytes(P, int(R)) if FULL_BACKWARD_SCRATCH else lib.gof_backward_scratch_bytes_for(P, int(R), staged)
            scratch = v.bytes_tensor(nscratch) if nscratch else None
            call = (v.ref(), int(R), _ptr(radii), _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer),
                    binningBuffer.numel(), _ptr(imageBuffer), imageBuffer.numel(), _ptr(dl),
                    _ptr(g_means2D), _ptr(g_colors), _ptr(g_opacity), _ptr(g_means3D), _ptr(g_cov3D) if want_cov3D else None,
                    _ptr(g_sh[0] if v.split_sh else g_sh), _ptr(g_sh[1]) if v.split_sh else None,
                    _ptr(g_scales), _ptr(g_rot), _ptr(g_v2g), _ptr(scratch), nscratch, _stream())
            track = _sh_track["on"] and M > 0
            src = None
            if track:
                _sh_track["count"] += 1
                src = _sh_track["src"] = {"dL_dcolors": g_colors, "geom": geomBuffer, "radii": radii, "means3D": v.keep["means3D"],
                                          "campos": v.keep["campos"], "degree": int(degree), "M": M, "P": P}
            early = track and _sh_track["ready_cb"] is not None
            # (a data-parallel reducer takes the colour gradient right after the blend stage, while preprocess_bwd is still to run
            # (dp/reducer.py): only the blend stage is queued here, the rest follows behind the verification below)
            _check((lib.gof_backward_blend if early else lib.gof_backward)(*call))
            if verify is not None:
                (words, ev), rec_guess = verify
                ev.synchronize()                      # the FORWARD's counters (long there: the GPU is busy with the backward just queued)
                q = (C.c_uint32 * 3)()
                _check(lib.gof_usage_decode(C.c_void_p(words.data_ptr()), int(R), W, H, binningBuffer.numel(), q))
                staged, requested, held = int(q[0]), int(q[1]), int(q[2])
                _mask_need[shape_key] = max(_mask_need.get(shape_key, 0), requested)
                _staged_need[shape_key] = max(_staged_need.get(shape_key, 0), staged)
                if requested > held:                  # masks are missing: the gradients just computed are incomplete -> forward again, then backward
                    if track:
                        # the abandoned backward must not count as a rasterizer backward of this step: the repeated one would make it
                        # two on THIS rank only, take_sh_grad_source() would return None here and the source on the other ranks, and the
                        # ranks would enter different collectives (dense all-reduce vs compressed all-gather)
                        _sh_track["count"] -= 1
                        _sh_track["src"] = None
                    raise MaskPoolTooSmall(requested, held)
                if staged > rec_guess:                # records were dropped: the same backward (stage) again, with room for all of them
                    _stats["record_pool_redone_backwards"] += 1
                    nscratch = lib.gof_backward_scratch_bytes_for(P, int(R), staged)
                    scratch = v.bytes_tensor(nscratch)
                    call = call[:-3] + (_ptr(scratch), nscratch, _stream())
                    _check((lib.gof_backward_blend if early else lib.gof_backward)(*call))
            if early:
                # the colour gradient is final and verified: the reducer starts its exchange, preprocess_bwd runs beside it
                _sh_track["ready_cb"](src)
                _check(lib.gof_backward_preprocess(*call))
    return g_means2D, g_colors, g_opacity, g_means3D, g_cov3D, g_sh, g_scales, g_rot, g_v2g


# ---- data-parallel training: the SH gradient in compressed form (include/gof_hip.h: gof_sh_grad_pack / gof_sh_grad_expand) ----
_sh_track = {"on": False, "count": 0, "src": None, "ready_cb": None}


def track_sh_grad_source(on=True):
    """While on, every backward remembers what the SH gradient of that view was expanded from (the blend's colour gradient, the
    forward's clamp flags, the camera centre).  dp/reducer.py enables it; nothing is kept otherwise."""
    _sh_track["on"] = bool(on)
    _sh_track["count"], _sh_track["src"] = 0, None
    if not on:
        _sh_track["ready_cb"] = None


def set_sh_grad_ready_callback(fn):
    """fn(src) is called inside every tracked backward right after the blend stage (the colour gradient of the view is final, the
    parameter gradients are not yet computed); None removes it."""
    _sh_track["ready_cb"] = fn


def take_sh_grad_source():
    """The source of the SH gradient accumulated since the last call, if it came from EXACTLY ONE rasterizer backward (the
    reference's training iteration, train.py:147-189); None otherwise (the caller then exchanges the dense gradient)."""
    n, src = _sh_track["count"], _sh_track["src"]
    _sh_track["count"], _sh_track["src"] = 0, None
    return src if n == 1 else None


def sh_grad_pack(src, out):
    """out[:P] (float32 [>=P,3], contiguous) = this view's masked colour gradient."""
    with torch.cuda.device(out.device):
        _check(lib.gof_sh_grad_pack(src["P"], _ptr(src["dL_dcolors"]), _ptr(src["geom"]), src["geom"].numel(), _ptr(src["radii"]),
                                    _ptr(out), _stream()))


def sh_grad_expand(src, gathered, scale, outs):
    """gathered: float32 [n_views, P+1, 3] (rows 0..P-1 of a view = its packed gradient, row P = its camera centre).
    outs: (dL_dsh [P,M,3],) or (dL_dfeatures_dc [P,1,3], dL_dfeatures_rest [P,M-1,3]); every element is overwritten."""
    P, M = src["P"], src["M"]
    n_views = int(gathered.shape[0])
    vs = (P + 1) * 3
    if len(outs) == 1:
        dc, rest = outs[0], outs[0]
        p_dc, p_rest, s_dc, s_rest = dc.data_ptr(), dc.data_ptr() + 12, 3 * M, 3 * M
    else:
        dc, rest = outs
        p_dc, p_rest, s_dc, s_rest = dc.data_ptr(), rest.data_ptr(), 3, 3 * (M - 1)
    if not (dc.is_contiguous() and rest.is_contiguous() and gathered.is_contiguous()):
        raise RuntimeError("sh_grad_expand: contiguous tensors expected")
    if dc.numel() + (rest.numel() if rest is not dc else 0) != 3 * M * P:
        raise RuntimeError("sh_grad_expand: the output tensors do not hold %d x %d x 3 floats" % (P, M))
    with torch.cuda.device(gathered.device):
        _check(lib.gof_sh_grad_expand(P, src["degree"], M, n_views, _ptr(src["means3D"]), gathered.data_ptr() + 12 * P, vs,
                                      gathered.data_ptr(), vs, float(scale), p_dc, s_dc, p_rest if M > 1 else None, s_rest, _stream()))


class IntegrateViewCache:
    """Per-view state of `integrate` that depends on the Gaussians and the camera only (records, sorted tile lists,
    per-pixel contributor masks, base image), kept across the 9-10 calls a mesh extraction makes per view with different
    query points (reference extract_mesh.py:23-31, 88-100).  Nothing is cached unless a driver
    announces a key with `integrate_view_key(key)`: the key is the driver's promise that Gaussians, camera and settings
    are unchanged (launch/run_reference_script.py derives it from the tensors' storage + version counters).
    Byte-budgeted, first come first kept (the access pattern is cyclic over the views, LRU would thrash)."""

    def __init__(self, max_bytes=None):
        self.max_bytes = max_bytes
        self.entries = {}
        self.bytes = 0
        self.hits = self.misses = self.rejected = 0

    def _budget(self, device):
        if self.max_bytes is None:
            env = os.environ.get("GOF_INTEGRATE_CACHE_GB")
            if env is not None:
                self.max_bytes = int(float(env) * (1 << 30))
            else:
                self.max_bytes = int(0.4 * torch.cuda.get_device_properties(device).total_memory)   # 115 GB of the 288 GB
        return self.max_bytes

    def get(self, key):
        e = self.entries.get(key)
        if e is None:
            self.misses += 1
        else:
            self.hits += 1
        return e

    def put(self, key, entry):
        nbytes = sum(t.numel() * t.element_size() for t in entry if isinstance(t, torch.Tensor))
        dev = next(t.device for t in entry if isinstance(t, torch.Tensor))
        free, total = torch.cuda.mem_get_info(dev)
        if self.bytes + nbytes > self._budget(dev) or free < 0.15 * total:      # never squeeze the caller's own working set
            self.rejected += 1
            return False
        self.entries[key] = entry
        self.bytes += nbytes
        return True

    def clear(self):
        self.entries.clear()
        self.bytes = 0


_view_cache = IntegrateViewCache()
_integrate_key = threading.local()


class integrate_view_key:
    """Context manager: `with integrate_view_key(key): rasterizer.integrate(...)` -- see IntegrateViewCache."""

    def __init__(self, key):
        self.key = key

    def __enter__(self):
        self.prev = getattr(_integrate_key, "value", None)
        _integrate_key.value = self.key
        return self

    def __exit__(self, *exc):
        _integrate_key.value = self.prev
        return False


def integrate_view_cache():
    return _view_cache


_integrate_acc = threading.local()


class integrate_min_into:
    """Context manager: `with integrate_min_into(alpha_min, color=None): rasterizer.integrate(points, ...)` -- the call min-combines its
    alpha_integrated into `alpha_min` [N] (float32, filled with 1 before the first view) in the kernel's final store and, where it
    lowers the minimum, writes the view's colour into `color` [N, 3]: the reduction over views of reference extract_mesh.py:17-34
    (`final_color = where(alpha < final_alpha, color, final_color); final_alpha = min(final_alpha, alpha)`) without the per-view
    `ones` / `zeros` fills and the separate min / where passes.  The call returns the two buffers as alpha_integrated /
    color_integrated and **None as its image**: the point pass writes no image in this mode (channel 8, the per-pixel point count,
    is not produced; gof_integrate_points_min)."""

    def __init__(self, alpha_min, color=None):
        self.value = (alpha_min, color)

    def __enter__(self):
        self.prev = getattr(_integrate_acc, "value", None)
        _integrate_acc.value = self.value
        return self

    def __exit__(self, *exc):
        _integrate_acc.value = self.prev
        return False


def integrate_gaussians_to_points(background, points3D, means3D, colors, opacity, scales, rotations, scale_modifier,
                                  cov3D_precomp, view2gaussian_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                                  kernel_size, subpixel_offset, image_height, image_width, sh, degree, campos,
                                  prefiltered, debug):
    """Replaces ``_C.integrate_gaussians_to_points`` (IntegrateGaussiansToPointsCUDA, rasterize_points.cu:234-343).
    Returns ``(num_rendered, out_color, out_alpha_integrated, out_color_integrated, radii, geomBuffer,
    binningBuffer, imgBuffer)``."""
    if points3D.dim() != 2 or points3D.size(1) != 3:
        raise RuntimeError("points3D must have dimensions (num_points, 3)")
    v = _View(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, view2gaussian_precomp,
              viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, subpixel_offset, image_height, image_width, sh,
              degree, campos, prefiltered, debug)
    dev = v.device
    PN = int(points3D.size(0))
    pts = _dev_f32(points3D, dev, "points3D")
    f = dict(dtype=torch.float32, device=dev)
    acc = getattr(_integrate_acc, "value", None)
    if acc is not None:
        acc_alpha, acc_color = acc
        if acc_alpha.dtype != torch.float32 or acc_alpha.device != dev or not acc_alpha.is_contiguous() or acc_alpha.numel() != PN:
            raise RuntimeError("integrate_min_into: alpha buffer must be a contiguous float32 tensor of %d elements on %s" % (PN, dev))
        if acc_color is not None and (acc_color.dtype != torch.float32 or acc_color.device != dev or not acc_color.is_contiguous()
                                      or tuple(acc_color.shape) != (PN, 3)):
            raise RuntimeError("integrate_min_into: colour buffer must be a contiguous float32 [%d, 3] tensor on %s" % (PN, dev))
    with torch.cuda.device(dev):
        out_color = torch.zeros((OUTPUT_CHANNELS, v.H, v.W), **f)
        if acc is None:
            out_alpha = torch.ones((PN,), **f)            # rasterize_points.cu:277
            out_color_pts = torch.zeros((PN, 3), **f)     # rasterize_points.cu:278
        else:
            out_alpha, out_color_pts = acc                # nothing to fill: the running buffers are updated in place
        radii = torch.zeros(v.P, dtype=torch.int32, device=dev)
        empty = v.bytes_tensor(0)
        if v.P == 0 or PN == 0:
            return 0, out_color, out_alpha, out_color_pts, radii, empty, empty.clone(), empty.clone()
        # Gaussian side of the view (binning + pixel pass): once per view key when a driver announces one (IntegrateViewCache)
        key = getattr(_integrate_key, "value", None)
        entry = _view_cache.get(key) if key is not None else None
        points_fn = lib.gof_integrate_points
        if entry is None:
            geom, img, binning, radii, rendered = _prepare_and_bin(v, for_query=True)
            base = out_color
            _check(lib.gof_integrate_view(v.ref(), rendered, _ptr(radii), _ptr(geom), geom.numel(), _ptr(binning), binning.numel(),
                                          _ptr(img), img.numel(), _ptr(base), _stream()))
            if key is not None:
                # keep only what the point pass reads of the geometry workspace (records + front depths: 68 of ~220 B per Gaussian)
                packed = v.bytes_tensor(lib.gof_integrate_packed_geom_bytes(v.P))
                _check(lib.gof_integrate_pack_geom(v.ref(), _ptr(geom), geom.numel(), _ptr(packed), packed.numel(), _stream()))
                if _view_cache.put(key, (packed, img, binning, radii, rendered, base, v.P, v.W, v.H)) and acc is None:
                    out_color = torch.empty_like(base)     # the cached base image must stay untouched by channel 8
        else:
            geom, img, binning, radii, rendered, base, cP, cW, cH = entry
            if (cP, cW, cH) != (v.P, v.W, v.H):
                raise RuntimeError("integrate view cache: key %r was announced for a different problem size" % (key,))
            if acc is None:
                out_color = torch.empty_like(base)
            points_fn = lib.gof_integrate_points_packed
        pws = v.bytes_tensor(lib.gof_point_bytes(PN))
        ni = C.c_uint32(0)
        _check(lib.gof_integrate_prepare_points(v.ref(), PN, _ptr(pts), _ptr(pws), pws.numel(), C.byref(ni), _stream()))
        pbin = v.bytes_tensor(lib.gof_point_binning_bytes(int(ni.value), v.W, v.H))
        if acc is not None:
            # min-accumulating call (integrate_min_into): the running minimum / arg-min colour buffers of the driver are the outputs
            acc_alpha, acc_color = acc
            _check(lib.gof_integrate_points_min(v.ref(), rendered, PN, int(ni.value), 1 if points_fn is lib.gof_integrate_points_packed else 0,
                                                _ptr(geom), geom.numel(), _ptr(binning), binning.numel(), _ptr(img), img.numel(),
                                                _ptr(pws), pws.numel(), _ptr(pbin), pbin.numel(), _ptr(base), None,
                                                _ptr(acc_alpha), None if acc_color is None else _ptr(acc_color), _stream()))
            # No image in this mode: the point pass writes none (channel 8, the per-pixel point count, is not produced), and a copy
            # of the view's cached base image per call was 9*H*W*4 bytes (61 MB at 1600x1063) nobody read -- the first element of
            # the returned tuple is None (documented on integrate_min_into).
            return rendered, None, acc_alpha, acc_color, radii, geom, binning, img
        _check(points_fn(v.ref(), rendered, PN, int(ni.value), _ptr(geom), geom.numel(), _ptr(binning), binning.numel(),
                                        _ptr(img), img.numel(), _ptr(pws), pws.numel(), _ptr(pbin), pbin.numel(), _ptr(base), _ptr(out_color),
                                        _ptr(out_alpha), _ptr(out_color_pts), _stream()))
    return rendered, out_color, out_alpha, out_color_pts, radii, geom, binning, img


def mark_visible(means3D, viewmatrix, projmatrix):
    """Replaces ``_C.mark_visible`` (rasterize_points.cu:213-232)."""
    dev = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros(P, dtype=torch.bool, device=dev)
    if P:
        m = _dev_f32(means3D, dev, "means3D"); vm = _dev_f32(viewmatrix, dev, "viewmatrix"); pm = _dev_f32(projmatrix, dev, "projmatrix")
        with torch.cuda.device(dev):
            _check(lib.gof_mark_visible(P, _ptr(m), _ptr(vm), _ptr(pm), C.c_void_p(present.data_ptr()), _stream()))
    return present


_FETCH = {"depths": (torch.float32, 1), "means2D": (torch.float32, 2), "conic_opacity": (torch.float32, 4), "rgb": (torch.float32, 3),
          "view2gaussian": (torch.float32, 10), "tiles_touched": (torch.int32, 1),
          "clamped": (torch.uint8, 3), "point_list": (torch.int32, 0), "point_list_keys": (torch.int64, 0),
          "ranges": (torch.int32, 0), "point_ranges": (torch.int32, 0), "final_T": (torch.float32, 0), "n_contrib": (torch.int32, 0),
          "contrib_pairs": (torch.int32, 0), "contrib_hash": (torch.int32, 0), "tile_cost": (torch.int32, 0), "tile_order": (torch.int32, 0), "tile_order_bw": (torch.int32, 0), "tile_queue": (torch.int32, 0)}


def debug_fetch(name, view, num_rendered, geom, binning, img):
    """Test/benchmark helper: copy a named intermediate out of the opaque workspaces (gof_debug_fetch)."""
    dtype, per = _FETCH[name]
    P, HW = view.P, view.H * view.W
    T = ((view.W + 15) // 16) * ((view.H + 15) // 16)
    count = {"point_list": _layout_count(num_rendered), "point_list_keys": _layout_count(num_rendered), "ranges": 2 * T, "point_ranges": 2 * T,
             "final_T": 4 * HW, "n_contrib": 2 * HW, "contrib_pairs": T, "contrib_hash": T, "tile_cost": T, "tile_order": 8 * ((T + 7) // 8 + 128), "tile_order_bw": 8 * ((T + 7) // 8 + 128), "tile_queue": 64}.get(name, P * per)
    out = torch.empty(count, dtype=dtype, device=view.device)
    n = lib.gof_debug_fetch(name.encode(), view.ref(), _layout_count(num_rendered), _ptr(geom), _ptr(binning), _ptr(img),
                            C.c_void_p(out.data_ptr()), out.numel() * out.element_size(), _stream())
    if n < 0:
        _check(int(n))
    return out


def set_forward_exact(on):
    """Verification mode of the forward blend (gof_set_forward_exact, include/gof_hip.h): True = every (pixel, Gaussian) pair in the
    reference's own arithmetic (every output bit the oracle's); False (default) = the same arithmetic without its two fp64 divisions per pair (pair_nodiv_cc: decisions and channels 0-7 identical on every scene tested).  The process-wide
    DEFAULT (calls inside a `call_modes(forward_exact=...)` block carry their own mode); returns the previous setting."""
    return bool(lib.gof_set_forward_exact(1 if on else 0))


def set_integrate_pixel_pass(on):
    """Pixel pass of the opacity-field query (gof_set_integrate_pixel_pass, include/gof_hip.h): False (default) = ray-centric, every distinct
    sub-ray of a tile once; True = pixel-centric (rounds 1-4).  Same outputs bit for bit.  Returns the previous setting."""
    return bool(lib.gof_set_integrate_pixel_pass(1 if on else 0))


def set_tight_tile_rects(on):
    """Opt-in tile lists (gof_set_tight_tile_rects, include/gof_hip.h): a Gaussian's tile rectangle intersected with its footprint
    box -- same image and gradients from shorter lists, which are then no longer the reference's entry for entry.  Process-wide;
    returns the previous setting."""
    return bool(lib.gof_set_tight_tile_rects(1 if on else 0))


def profile_enable(on=True):
    """Bracket every kernel launch with HIP events on the launch stream (gof_profile_enable)."""
    _check(lib.gof_profile_enable(1 if on else 0))


def profile_report():
    """-> {kernel: {"calls": n, "total_ms": t}}; waits for the recorded events and clears them."""
    import json
    buf = C.create_string_buffer(1 << 16)
    _check(lib.gof_profile_report(buf, len(buf)))
    return json.loads(buf.value.decode())
