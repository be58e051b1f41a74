"""Training-iteration epilogue on gfx950: drop-in mirrors of the reference's pure-torch
``utils.loss_utils.ssim``, ``utils.depth_utils.depth_to_normal`` and of the ``torch.optim.Adam`` instance that
``scene/gaussian_model.py:360`` builds.  launch/run_reference_script.py rebinds them into the unchanged train.py."""
from .loss_utils import ssim, l1_loss, l2_loss          # noqa: F401
from .fused_loss import training_loss, TrainingLoss         # noqa: F401
from .depth_utils import depth_to_normal, depths_to_points  # noqa: F401
from .optim import FusedAdam                             # noqa: F401
from .filter_3d import compute_3D_filter, filter_3d, camera_table, add_densification_stats   # noqa: F401
from . import activations                                # noqa: F401,E402
from .densify import densify_and_prune                  # noqa: F401,E402
from .pose import PoseMatrix, SmallMatrix                # noqa: F401,E402
