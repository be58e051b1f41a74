"""Per-view data parallelism for Gaussian training (no reference counterpart: the reference is
single-GPU).  One process per GPU, a full replica of the Gaussians on each, one
camera per rank per step, one exchange per optimiser step: the parameter gradients (59 floats = 236 B
per Gaussian) over RCCL/xGMI -- a dense all-reduce, or the SH gradient as 12 B per Gaussian and view
(all-gather + local expansion, gof_sh_grad_pack / gof_sh_grad_expand) and an all-reduce of the other 44 B."""
from .reducer import GradientAllReducer, ViewShards, shard_views  # noqa: F401
from .mesh import evaluate_alpha  # noqa: F401,E402
