"""m4depth_amd -- MI355X-native implementation of M4Depth's per-frame
parallax-cost-volume inference path (hand-written HIP for gfx950 behind the
reference's Python layer/op API).  Importing the package loads
libm4depth_hip.so and raises if it is missing: there is no CPU fallback.
"""
from . import _lib                                   # noqa: F401  (fails loudly without the native library)
from .dense_image_warp import dense_image_warp, back_project, back_project_grad, _interpolate_bilinear  # noqa: F401
from .depth_operations import (get_rot_mat, get_coords_2d, reproject, recompute_depth, parallax2depth,  # noqa: F401
                               depth2parallax, prev_d2para, tile_in_batch, get_parallax_sweeping_cv,
                               cost_volume, wrap_feature_block)
from .network import (M4Depth, M4depthAblationParameters, DepthEstimatorPyramid, DepthEstimatorLevel,    # noqa: F401
                      FeaturePyramid, DispRefiner, DomainNormalization)
from .metrics import (RootMeanSquaredError, RootMeanSquaredLogError, AbsRelError, SqRelError,            # noqa: F401
                      ThresholdRelError, default_metrics)

__version__ = "0.1.0"
