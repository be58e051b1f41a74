"""Synthetic scenes for parity tests and benchmarks (no dataset is available offline).

Recipes follow SURVEY.md section 8(d) / BASELINE.md section 2:

* ``scene_frustum`` ("S1M" at P=1_000_000, 1600x1063): camera at the origin looking +z,
  Gaussians fill the frustum with 10 % overscan, sigma_px ~ LogNormal(ln 3, 0.5), per-axis
  anisotropy LogNormal(0, 0.5), random unit quaternions, opacity ~ U[0.1, 0.9], SH degree 3.
* ``scene_lego_like`` ("S10k", BASELINE config 1): P=10_000 points U[-1.3,1.3]^3 (as
  reference scene/dataset_readers.py:241-247), camera on a radius-4.03 sphere looking at the
  origin, camera_angle_x = 0.6911 (focal ~555.6 px at 400 px), white background.

Camera matrices use the reference conventions (utils/graphics_utils.py:38-71,
scene/cameras.py:50-58): ``viewmatrix`` = world-to-view, TRANSPOSED (row-vector
convention); ``projmatrix`` = viewmatrix @ projection^T.

Everything is generated on the CPU with numpy (seeded) and returned as a dict of float32
numpy arrays + python scalars; callers move them to the device.
"""
import math
import numpy as np


def projection_matrix(znear, zfar, fovx, fovy):
    """utils/graphics_utils.py:51-71 (returns the NON-transposed 4x4, float32)."""
    tan_y = math.tan(fovy / 2)
    tan_x = math.tan(fovx / 2)
    top = tan_y * znear
    bottom = -top
    right = tan_x * znear
    left = -right
    P = np.zeros((4, 4), dtype=np.float32)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera(W, H, fovx, fovy, R=None, T=None, znear=0.01, zfar=100.0):
    """R: camera-to-world rotation (3x3), T: world-to-view translation, as scene/cameras.py."""
    if R is None:
        R = np.eye(3)
    if T is None:
        T = np.zeros(3)
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = T
    Rt[3, 3] = 1.0
    w2v = np.float32(Rt)                      # getWorld2View2 with zero translate, scale 1
    view_t = np.ascontiguousarray(w2v.T)      # .transpose(0,1)
    proj_t = np.ascontiguousarray(projection_matrix(znear, zfar, fovx, fovy).T)
    full = (view_t @ proj_t).astype(np.float32)
    campos = np.linalg.inv(view_t.astype(np.float64))[3, :3].astype(np.float32)
    return dict(W=int(W), H=int(H), tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5),
                viewmatrix=view_t, projmatrix=np.ascontiguousarray(full), campos=campos)


def _rand_sh(rng, P, deg_max=3):
    M = (deg_max + 1) ** 2
    sh = np.empty((P, M, 3), dtype=np.float32)
    sh[:, 0] = rng.normal(0.0, 0.5, (P, 3))
    if M > 1:
        sh[:, 1:] = rng.normal(0.0, 0.1, (P, M - 1, 3))
    return sh


def scene_frustum(P, W=1600, H=1063, focal=1200.0, seed=0, sh_degree=3, sigma_px=3.0,
                  zmin=1.0, zmax=20.0, bg=(0.0, 0.0, 0.0), kernel_size=0.0, pose_seed=None):
    """pose_seed: None = the SURVEY recipe (camera at the origin, viewmatrix = I); an int = the same cloud under pose_scene()."""
    rng = np.random.default_rng(seed)
    tanx = (W / 2) / focal
    tany = (H / 2) / focal
    fovx = 2 * math.atan(tanx)
    fovy = 2 * math.atan(tany)
    cam = camera(W, H, fovx, fovy)
    z = rng.uniform(zmin, zmax, P)
    x = z * tanx * rng.uniform(-1.1, 1.1, P)
    y = z * tany * rng.uniform(-1.1, 1.1, P)
    means = np.stack([x, y, z], 1).astype(np.float32)
    base = np.exp(rng.normal(math.log(sigma_px), 0.5, P))          # pixels
    aniso = np.exp(rng.normal(0.0, 0.5, (P, 3)))
    scales = (base[:, None] * z[:, None] / focal * aniso).astype(np.float32)
    q = rng.normal(0, 1, (P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    rot = q.astype(np.float32)
    opac = rng.uniform(0.1, 0.9, (P, 1)).astype(np.float32)
    sh = _rand_sh(rng, P)
    scene = dict(cam)
    scene.update(means3D=means, scales=scales, rotations=rot, opacities=opac, shs=sh,
                 sh_degree=int(sh_degree), bg=np.asarray(bg, dtype=np.float32), kernel_size=float(kernel_size),
                 scale_modifier=1.0, subpixel_offset=np.zeros((H, W, 2), dtype=np.float32))
    return scene if pose_seed is None else pose_scene(scene, pose_seed)


def scene_clustered(P=1_000_000, W=1600, H=1063, focal=1200.0, seed=0, sh_degree=3, kernel_size=0.0, pose_seed=None,
                    frac_clustered=0.70, frac_large=0.02, n_clusters=5):
    """"S1M-clustered": a heavy-tailed sibling of scene_frustum for the tile scheduler (real captures have empty sky tiles next to
    10k-entry foreground tiles; S1M's lists are 1319 +- 100 entries and every tile costs the same).  70 % of the Gaussians sit in
    `n_clusters` compact blobs that together fill ~5 % of the frustum volume (Gaussian balls in (x/z, y/z, log z)) and are
    semi-transparent (opacity U[0.02, 0.3]: foliage-like, a pixel blends hundreds of them before it saturates); 2 % are large
    splats (sigma_px 30-100) at the back of the scene (depth 15-20: walls / sky, they lengthen every list without saturating
    the front); the remaining 28 % are scene_frustum's uniform background.  Anisotropy, quaternions, SH follow scene_frustum."""
    rng = np.random.default_rng(seed + 77)
    sc = scene_frustum(P, W=W, H=H, focal=focal, seed=seed, sh_degree=sh_degree, kernel_size=kernel_size)
    tanx, tany = sc["tanfovx"], sc["tanfovy"]
    n_c = int(P * frac_clustered); n_l = int(P * frac_large)
    means = sc["means3D"]; scales = sc["scales"]
    cu = rng.uniform(-0.8, 0.8, (n_clusters, 2)); cz = rng.uniform(3.0, 12.0, n_clusters)
    which = rng.integers(0, n_clusters, n_c)
    s_rel = 0.105                                          # n blobs of relative radius s fill ~ n (4/3) pi s^3 / 8 of the (u, v, log z) box
    u = cu[which, 0] + rng.normal(0, s_rel, n_c)
    v = cu[which, 1] + rng.normal(0, s_rel, n_c)
    z = cz[which] * np.exp(rng.normal(0, s_rel * 1.5, n_c))
    idx = rng.permutation(P)
    ic, il = idx[:n_c], idx[n_c:n_c + n_l]
    means[ic] = np.stack([u * z * tanx, v * z * tany, z], 1).astype(np.float32)
    base = np.exp(rng.normal(math.log(2.0), 0.5, n_c))
    scales[ic] = (base[:, None] * z[:, None] / focal * np.exp(rng.normal(0.0, 0.5, (n_c, 3)))).astype(np.float32)
    sc["opacities"][ic] = rng.uniform(0.02, 0.3, (n_c, 1)).astype(np.float32)
    zl = rng.uniform(15.0, 20.0, n_l)
    means[il] = np.stack([zl * tanx * rng.uniform(-1.0, 1.0, n_l), zl * tany * rng.uniform(-1.0, 1.0, n_l), zl], 1).astype(np.float32)
    big = rng.uniform(30.0, 100.0, n_l)
    scales[il] = (big[:, None] * zl[:, None] / focal * np.exp(rng.normal(0.0, 0.3, (n_l, 3)))).astype(np.float32)
    return sc if pose_seed is None else pose_scene(sc, pose_seed)


def _quat_to_rot(q):
    r, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                     [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                     [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def _quat_mul(a, b):
    """Hamilton product a (x) b, (r, x, y, z) order as the reference's rotations; a: (4,), b: (P, 4)."""
    ar, ax, ay, az = a
    br, bx, by, bz = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return np.stack([ar * br - ax * bx - ay * by - az * bz,
                     ar * bx + ax * br + ay * bz - az * by,
                     ar * by - ax * bz + ay * br + az * bx,
                     ar * bz + ax * by - ay * bx + az * br], 1)


def pose_scene(scene, seed, spread=3.0):
    """Re-pose a scene built in the camera frame (viewmatrix = I, campos = 0: scene_frustum and what tests derive from it) under a
    random rigid motion: camera-to-world rotation R (uniform, from a random unit quaternion), camera centre c ~ N(0, spread^2).
    Means go to world = R x + c, every Gaussian's quaternion is composed with R's (so the splats keep their place and shape in
    the image up to fp32 rounding), and viewmatrix / projmatrix / campos are the matching world-to-view camera in the reference's
    conventions (scene/cameras.py:50-58) -- the configuration every real training step renders: a posed camera looking at
    rotated anisotropic Gaussians.  Returns a new dict; the input is not modified."""
    rng = np.random.default_rng(7000 + seed)
    qc = rng.normal(0, 1, 4)
    qc /= np.linalg.norm(qc)
    R = _quat_to_rot(qc)                                    # camera-to-world
    c = rng.normal(0.0, spread, 3)
    out = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in scene.items()}
    out["means3D"] = (scene["means3D"].astype(np.float64) @ R.T + c).astype(np.float32)
    q = _quat_mul(qc, scene["rotations"].astype(np.float64))
    nrm = np.linalg.norm(scene["rotations"].astype(np.float64), axis=1, keepdims=True)       # keeps un-normalised inputs un-normalised
    qn = np.linalg.norm(q, axis=1, keepdims=True)
    out["rotations"] = (q / np.where(qn > 0, qn, 1.0) * nrm).astype(np.float32)
    fovx, fovy = 2 * math.atan(scene["tanfovx"]), 2 * math.atan(scene["tanfovy"])
    out.update(camera(scene["W"], scene["H"], fovx, fovy, R=R, T=-R.T @ c))
    return out


def other_view(scene, k=1, angle=0.06, shift=0.12):
    """The SAME Gaussians seen by another camera: view k of a small orbit around the scene's own camera (camera-to-world rotation by
    k * angle about the y axis composed with the scene's, centre moved by k * shift along the camera's x and -y axes) -- what a
    data-parallel step renders on rank k.  Only viewmatrix / projmatrix / campos change; most of the cloud stays in view."""
    V = scene["viewmatrix"].astype(np.float64)               # world-to-view, transposed (row-vector convention)
    R0 = V[:3, :3]                                           # = camera-to-world rotation (view_t[:3,:3] = R)
    c0 = np.linalg.inv(V)[3, :3]
    a = k * angle
    Ry = np.array([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
    R = R0 @ Ry
    c = c0 + R0 @ np.array([k * shift, -0.5 * k * shift, 0.0])
    out = {key: (v.copy() if isinstance(v, np.ndarray) else v) for key, v in scene.items()}
    fovx, fovy = 2 * math.atan(scene["tanfovx"]), 2 * math.atan(scene["tanfovy"])
    out.update(camera(scene["W"], scene["H"], fovx, fovy, R=R, T=-R.T @ c))
    return out


def scene_lego_like(P=10_000, W=400, H=400, seed=0, sh_degree=3, bg=(1.0, 1.0, 1.0), kernel_size=0.0):
    rng = np.random.default_rng(seed)
    fovx = 0.6911112070083618
    focal = W / (2 * math.tan(fovx / 2))
    fovy = 2 * math.atan(H / (2 * focal))
    # camera on a sphere of radius 4.03 looking at the origin
    theta, phi = 0.7, 0.5
    c = 4.03 * np.array([math.cos(phi) * math.sin(theta), -math.sin(phi), math.cos(phi) * math.cos(theta)])
    fwd = -c / np.linalg.norm(c)
    up = np.array([0.0, -1.0, 0.0])
    right = np.cross(up, fwd); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R_c2w = np.stack([right, down, fwd], 1)       # columns = camera axes in world
    T = -R_c2w.T @ c
    cam = camera(W, H, fovx, fovy, R=R_c2w, T=T)
    means = rng.uniform(-1.3, 1.3, (P, 3)).astype(np.float32)
    # isotropic scale from the mean distance to the 3 nearest neighbours (simple_knn.distCUDA2 stand-in)
    try:
        from scipy.spatial import cKDTree
        d, _ = cKDTree(means).query(means, k=4)
        dist2 = np.maximum((d[:, 1:] ** 2).mean(1), 1e-7)
    except Exception:  # pragma: no cover
        dist2 = np.full(P, 0.01)
    scales = np.repeat(np.sqrt(dist2)[:, None], 3, 1).astype(np.float32)
    rot = np.zeros((P, 4), dtype=np.float32); rot[:, 0] = 1.0
    opac = np.full((P, 1), 0.1, dtype=np.float32)
    sh = _rand_sh(rng, P)
    scene = dict(cam)
    scene.update(means3D=means, scales=scales, rotations=rot, opacities=opac, shs=sh,
                 sh_degree=int(sh_degree), bg=np.asarray(bg, dtype=np.float32), kernel_size=float(kernel_size),
                 scale_modifier=1.0, subpixel_offset=np.zeros((H, W, 2), dtype=np.float32))
    return scene


def tetra_points(scene, per_gaussian=9):
    """Query points as GaussianModel.get_tetra_points builds them (scene/gaussian_model.py:432-463):
    8 corners of the 3-sigma box + the centre, without the frustum mask."""
    means, scales, rot = scene["means3D"], scene["scales"] * 3.0, scene["rotations"]
    r, x, y, z = rot[:, 0], rot[:, 1], rot[:, 2], rot[:, 3]
    R = np.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float32)
    v = corners[None] * scales[:, None, :]                 # (P,8,3)
    v = np.einsum("pij,pkj->pki", R, v) + means[:, None, :]
    pts = np.concatenate([v.reshape(-1, 3), means], 0).astype(np.float32)
    return np.ascontiguousarray(pts)


def freudenthal_tets(nx, ny, nz):
    """6-tets-per-cube grid tetrahedralisation (CGAL Delaunay stand-in). Returns (verts (V,3) f32, tets (T,4) i64)."""
    xs, ys, zs = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), np.arange(nz + 1), indexing="ij")
    verts = np.stack([xs, ys, zs], -1).reshape(-1, 3).astype(np.float32)

    def vid(i, j, k):
        return (i * (ny + 1) + j) * (nz + 1) + k
    ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    ii, jj, kk = ii.ravel(), jj.ravel(), kk.ravel()
    c = [vid(ii + a, jj + b, kk + d) for a in (0, 1) for b in (0, 1) for d in (0, 1)]  # index = a*4+b*2+d
    perms = [(4, 6), (4, 5), (2, 6), (2, 3), (1, 5), (1, 3)]
    tets = [np.stack([c[0], c[p], c[q], c[7]], 1) for p, q in perms]
    return verts, np.ascontiguousarray(np.concatenate(tets, 0).astype(np.int64))
