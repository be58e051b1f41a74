cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_train_epilogue_gpu.py -m gpu -q -k "activations" 2>&1 | grep -E "passed|failed|^E  " | cut -c1-300 | head
python - <<'PY'
import sys, time, types, torch
sys.path.insert(0, "gaussian-opacity-fields_amd"); sys.path.insert(0, "oracle")
import train_epilogue as T, train_epilogue_oracle as O
P = 1_000_000
m = types.SimpleNamespace(_scaling=(torch.randn(P, 3, device="cuda") - 4).requires_grad_(True), _opacity=torch.randn(P, 1, device="cuda").requires_grad_(True),
                          _rotation=torch.randn(P, 4, device="cuda").requires_grad_(True), filter_3D=torch.rand(P, 1, device="cuda") * 0.05)
ws, wo, wr = torch.randn(P, 3, device="cuda"), torch.randn(P, 1, device="cuda"), torch.randn(P, 4, device="cuda")
def run(fs, fo, fr):
    for t in (m._scaling, m._opacity, m._rotation): t.grad = None
    ((fs() * ws).sum() + (fo() * wo).sum() + (fr() * wr).sum()).backward()
A = T.activations
hip = lambda: run(lambda: A.get_scaling_with_3D_filter(m), lambda: A.get_opacity_with_3D_filter(m), lambda: A.get_rotation(m))
ref = lambda: run(lambda: O.scaling_with_3D_filter(m._scaling, m.filter_3D), lambda: O.opacity_with_3D_filter(m._opacity, m._scaling, m.filter_3D), lambda: O.rotation(m._rotation))
for name, fn in (("hip", hip), ("torch (reference code)", ref)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print(name, "activations fwd+bwd (incl. the test's 3 weighted sums): %.3f ms" % ((time.perf_counter() - t) / 20 * 1e3))
PY
