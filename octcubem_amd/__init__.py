"""octcubem_amd -- MI355X-native (gfx950) 3-D MAE hot path of OCTCubeM.

Only what the path needs: HIP kernels + C ABI (csrc/, liboctmae.so), and the host-side mirror of the
reference's Python interface for this path (video_vit, models_mae, misc, lr_sched, engine_pretrain).
"""
from . import _lib  # noqa: F401  (does not load the shared library until first use)

__all__ = ["models_mae", "video_vit", "misc", "lr_sched", "engine_pretrain", "optim", "parallel", "ops"]
