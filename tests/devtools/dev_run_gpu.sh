cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_train_epilogue_gpu.py -m gpu -q -k "3d_filter" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | cut -c1-250 | head
python - <<'PY'
import sys, time, types, torch, numpy as np
sys.path.insert(0, "gaussian-opacity-fields_amd"); sys.path.insert(0, "oracle")
import train_epilogue as T, train_epilogue_oracle as O
rng = np.random.default_rng(0)
P, ncam = 1_000_000, 300
xyz = torch.from_numpy(rng.uniform(-2, 2, (P, 3)).astype(np.float32)).cuda()
cams = [types.SimpleNamespace(R=np.eye(3), T=np.array([0, 0, 4.0 + 0.01 * i]), focal_x=1000.0, focal_y=1000.0, image_width=1600, image_height=1063) for i in range(ncam)]
tab = T.camera_table(cams, "cuda")
T.filter_3d(xyz, tab); torch.cuda.synchronize(); t = time.perf_counter(); T.filter_3d(xyz, tab); torch.cuda.synchronize()
print("HIP compute_3D_filter, 1M points x 300 cameras: %.2f ms" % ((time.perf_counter() - t) * 1e3))
# the reference's code path on the same GPU (torch ops per camera)
def ref(xyz, cameras):
    distance = torch.ones((xyz.shape[0]), device=xyz.device) * 100000.0
    valid_points = torch.zeros((xyz.shape[0]), device=xyz.device, dtype=torch.bool)
    focal_length = 0.
    for camera in cameras:
        R = torch.tensor(camera.R, device=xyz.device, dtype=torch.float32); Tt = torch.tensor(camera.T, device=xyz.device, dtype=torch.float32)
        xyz_cam = xyz @ R + Tt[None, :]
        valid_depth = xyz_cam[:, 2] > 0.2
        x, y, z = xyz_cam[:, 0], xyz_cam[:, 1], xyz_cam[:, 2]
        z = torch.clamp(z, min=0.001)
        x = x / z * camera.focal_x + camera.image_width / 2.0; y = y / z * camera.focal_y + camera.image_height / 2.0
        in_screen = torch.logical_and(torch.logical_and(x >= -0.15 * camera.image_width, x <= camera.image_width * 1.15), torch.logical_and(y >= -0.15 * camera.image_height, y <= 1.15 * camera.image_height))
        valid = torch.logical_and(valid_depth, in_screen)
        distance[valid] = torch.min(distance[valid], z[valid]); valid_points = torch.logical_or(valid_points, valid)
        if focal_length < camera.focal_x: focal_length = camera.focal_x
    distance[~valid_points] = distance[valid_points].max()
    return (distance / focal_length * (0.2 ** 0.5))[..., None]
ref(xyz, cams[:3]); torch.cuda.synchronize(); t = time.perf_counter(); r = ref(xyz, cams); torch.cuda.synchronize()
print("reference torch code on this GPU: %.1f ms" % ((time.perf_counter() - t) * 1e3))
PY
