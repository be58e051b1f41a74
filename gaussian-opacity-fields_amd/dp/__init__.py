"""Per-view data parallelism for Gaussian training (no reference counterpart: the reference is
single-GPU).  One process per GPU, a full replica of the Gaussians on each, one
camera per rank per step, one exchange per optimiser step: an all-reduce of the dense parameter
gradients (59 floats = 236 B per Gaussian) over RCCL/xGMI."""
from .reducer import GradientAllReducer, shard_views  # noqa: F401
from .mesh import evaluate_alpha  # noqa: F401,E402
