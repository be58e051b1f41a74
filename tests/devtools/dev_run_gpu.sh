cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_knn.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | cut -c1-250 | head -20
python - <<'PY'
import sys, time, torch, numpy as np
sys.path.insert(0, "gaussian-opacity-fields_amd")
from simple_knn._C import distCUDA2
for n in (100_000, 1_000_000, 5_000_000):
    p = torch.from_numpy(np.random.default_rng(0).uniform(-1.3, 1.3, (n, 3)).astype(np.float32)).cuda()
    distCUDA2(p); torch.cuda.synchronize(); t = time.perf_counter(); distCUDA2(p); torch.cuda.synchronize()
    print(n, "points: %.2f ms" % ((time.perf_counter() - t) * 1e3))
PY
