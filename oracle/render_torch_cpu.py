"""TEST INFRASTRUCTURE (never imported by the product): a dense PyTorch-CPU restatement of the GOF forward -- SURVEY.md 7 step 1's
`render_torch_cpu`, the CPU-runnable form of BASELINE config 1 (the reference itself has no CPU path: gaussian_renderer/__init__.py:26
hard-wires "cuda").  Independent of oracle/gof_oracle.cpp in structure and precision: float64, whole-image tensors, no tile lists, no
sort keys, no LDS-shaped loops -- the model of reference submodules/diff-gaussian-rasterization/cuda_rasterizer/forward.cu written as
mathematics:

  * per Gaussian (forward.cu:283-404): view-space centre, 3D covariance R S^2 R^T, EWA 2D covariance + the mip low-pass `kernel_size`
    (coef = sqrt(det / det_filtered)), 3-sigma radius and the tile rectangle it touches (auxiliary.h getRect), SH colour, and the
    ray-space quadratic q(x) = (x - mu)^T M (x - mu), M = Rv diag(1 / (s^2 + 1e-7)) Rv^T (computeView2Gaussian, :168-261);
  * per pixel ray r = ((px + .5 - W/2)/fx, (py + .5 - H/2)/fy, 1) (forward.cu:409-612): along the ray q = AA t^2 + BB t + CC with
    AA = r^T M r, BB = -2 r^T M mu, CC = mu^T M mu; the Gaussian acts at the ray's point of maximum density t* = -BB / (2 AA) with
    alpha = min(0.99, w exp(-q(t*) / 2)); skipped for t* <= 0.2 or alpha < 1/255; front-to-back blending in the order of the
    centres' view depth (ties: index), stopped when T (1 - alpha) < 1e-4;
  * outputs [9, H, W]: colour + T bg, sum of alpha T (-M r / |M r|) (normal), the depth t* of the last Gaussian blended while T > 0.5,
    sum of alpha T, and the distortion of the depths mapped to [0, 1] (m(t) = (100 t - 20) / (99.8 t)) normalised by (1 - T)^2 + 1e-7.

A Gaussian touches a pixel only inside the 16x16 tiles its rectangle covers -- part of the reference's semantics (alpha can still
exceed 1/255 at 3 sigma), so it is part of this restatement.  Agreement with the oracle is not bit for bit (different precision and
operation order); tests/test_oracle_pins.py holds the two to 2e-5 outside the handful of pixels where a threshold decision flips.
"""
import math

import numpy as np
import torch

NEAR, FAR = 0.2, 100.0
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


def _rotation(q):
    r, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def _sh_colour(deg, means, campos, sh):
    d = means - campos
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    c = SH_C0 * sh[:, 0]
    if deg > 0:
        c = c - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        c = (c + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        c = (c + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10] + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
             + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12] + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
             + SH_C3[5] * z * (xx - yy) * sh[:, 14] + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp(c + 0.5, min=0.0)


@torch.no_grad()
def render_torch_cpu(scene, dtype=torch.float64):
    """scene: the dict of synthetic_scenes (numpy).  Returns (image [9, H, W] float64 numpy, radii [P] int32 numpy)."""
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)       # noqa: E731
    W, H = int(scene["W"]), int(scene["H"])
    tanx, tany = float(scene["tanfovx"]), float(scene["tanfovy"])
    fx, fy = W / (2 * tanx), H / (2 * tany)
    ks, mod, deg = float(scene["kernel_size"]), float(scene["scale_modifier"]), int(scene["sh_degree"])
    VM, PM = t(scene["viewmatrix"]), t(scene["projmatrix"])         # row-vector convention: p_view = [p, 1] @ VM
    means, scales, quats = t(scene["means3D"]), t(scene["scales"]), t(scene["rotations"])
    opac, sh, bg, campos = t(scene["opacities"]).reshape(-1), t(scene["shs"]), t(scene["bg"]), t(scene["campos"])
    P = means.shape[0]
    gx, gy = (W + 15) // 16, (H + 15) // 16

    Wr = VM[:3, :3].T                                                # world -> view rotation
    mu = means @ VM[:3, :3] + VM[3, :3]                              # view-space centres
    Rg = _rotation(quats)
    # ---- footprint: 3D covariance, EWA projection, low-pass, radius, tile rectangle
    S = torch.diag_embed(mod * scales)
    Sig = Rg @ S @ S @ Rg.transpose(1, 2)
    limx, limy = 1.3 * tanx, 1.3 * tany
    tz = mu[:, 2]
    tx = torch.clamp(mu[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(mu[:, 1] / tz, -limy, limy) * tz
    J = torch.zeros(P, 2, 3, dtype=dtype)
    J[:, 0, 0] = fx / tz; J[:, 0, 2] = -fx * tx / (tz * tz)
    J[:, 1, 1] = fy / tz; J[:, 1, 2] = -fy * ty / (tz * tz)
    JW = J @ Wr
    cov = JW @ Sig @ JW.transpose(1, 2)
    c00, c01, c11 = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    det0 = torch.clamp(c00 * c11 - c01 * c01, min=1e-6)
    det1 = torch.clamp((c00 + ks) * (c11 + ks) - c01 * c01, min=1e-6)
    coef = torch.sqrt(det0 / (det1 + 1e-6) + 1e-6)
    coef = torch.where((det0 <= 1e-6) | (det1 <= 1e-6), torch.zeros_like(coef), coef)
    a, b, c = c00 + ks, c01, c11 + ks
    det = a * c - b * b
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam))
    hom = torch.cat([means, torch.ones(P, 1, dtype=dtype)], 1) @ PM
    pw = 1.0 / (hom[:, 3] + 1e-7)
    pix = ((hom[:, 0] * pw + 1.0) * W - 1.0) * 0.5
    piy = ((hom[:, 1] * pw + 1.0) * H - 1.0) * 0.5
    trunc = lambda v: torch.trunc(v).to(torch.int64)                # noqa: E731  C's (int) conversion
    minx = torch.clamp(trunc((pix - radius) / 16), 0, gx); maxx = torch.clamp(trunc((pix + radius + 15) / 16), 0, gx)
    miny = torch.clamp(trunc((piy - radius) / 16), 0, gy); maxy = torch.clamp(trunc((piy + radius + 15) / 16), 0, gy)
    visible = (tz > NEAR) & (det != 0) & ((maxx - minx) * (maxy - miny) > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    # ---- the quadratic along rays: M, b = -M mu, CC
    Rv = Wr @ Rg                                                     # Gaussian axes in view space
    D = torch.diag_embed(1.0 / (scales * scales + 1e-7))
    M = Rv @ D @ Rv.transpose(1, 2)
    bvec = -(M @ mu.unsqueeze(-1)).squeeze(-1)
    CC = (mu.unsqueeze(1) @ M @ mu.unsqueeze(-1)).reshape(-1)
    wgt = opac * coef
    colour = _sh_colour(deg, means, campos, sh)

    ys, xs = torch.meshgrid(torch.arange(H, dtype=dtype), torch.arange(W, dtype=dtype), indexing="ij")
    RX = (xs + 0.5 - W / 2.0) / fx
    RY = (ys + 0.5 - H / 2.0) / fy
    T = torch.ones(H, W, dtype=dtype)
    done = torch.zeros(H, W, dtype=torch.bool)
    out = torch.zeros(9, H, W, dtype=dtype)
    dist1 = torch.zeros(H, W, dtype=dtype); dist2 = torch.zeros(H, W, dtype=dtype); dist = torch.zeros(H, W, dtype=dtype)
    order = sorted(np.nonzero(visible.numpy())[0].tolist(), key=lambda i: (np.float32(tz[i].item()), i))   # fp32 depth key, ties by index
    for i in order:
        y0, y1 = int(miny[i]) * 16, min(H, int(maxy[i]) * 16)
        x0, x1 = int(minx[i]) * 16, min(W, int(maxx[i]) * 16)
        sl = (slice(y0, y1), slice(x0, x1))
        rx, ry = RX[sl], RY[sl]
        Mi = M[i]
        n0 = Mi[0, 0] * rx + Mi[0, 1] * ry + Mi[0, 2]
        n1 = Mi[1, 0] * rx + Mi[1, 1] * ry + Mi[1, 2]
        n2 = Mi[2, 0] * rx + Mi[2, 1] * ry + Mi[2, 2]
        AA = rx * n0 + ry * n1 + n2
        BB = 2 * (bvec[i, 0] * rx + bvec[i, 1] * ry + bvec[i, 2])
        ts = -BB / (2 * AA)
        power = torch.clamp(-0.5 * (CC[i] - BB * BB / (4 * AA)), max=0.0)
        alpha = torch.clamp(wgt[i] * torch.exp(power), max=0.99)
        Tl = T[sl]
        use = (~done[sl]) & (ts > NEAR) & (alpha >= 1.0 / 255.0)
        test_T = Tl * (1 - alpha)
        stop = use & (test_T < 1e-4)
        done[sl] |= stop
        use = use & ~stop
        if not bool(use.any()):
            continue
        aT = torch.where(use, alpha * Tl, torch.zeros_like(alpha))
        m = (FAR * ts - FAR * NEAR) / ((FAR - NEAR) * ts)
        A = 1 - Tl
        err = m * m * A + dist2[sl] - 2 * m * dist1[sl]
        dist[sl] += torch.where(use, err, torch.zeros_like(err)) * aT
        dist1[sl] += m * aT
        dist2[sl] += m * m * aT
        inv = -1.0 / torch.sqrt(n0 * n0 + n1 * n1 + n2 * n2 + 1e-7)
        for ch in range(3):
            out[ch][sl] += colour[i, ch] * aT
        out[3][sl] += n0 * inv * aT
        out[4][sl] += n1 * inv * aT
        out[5][sl] += n2 * inv * aT
        out[6][sl] = torch.where(use & (Tl > 0.5), ts, out[6][sl])
        out[7][sl] += aT
        T[sl] = torch.where(use, test_T, Tl)
    for ch in range(3):
        out[ch] += T * bg[ch]
    out[8] = dist / ((1 - T) * (1 - T) + 1e-7)
    return out.numpy(), radii.numpy()
