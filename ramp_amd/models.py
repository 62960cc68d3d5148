"""Namespace mirroring ``mpd.models`` for the sampler hot path."""
from .diffusion import (DynamicGaussianDiffusionModel, GaussianDiffusionModel3d,  # noqa: F401
                        StaticGaussianDiffusionModel)
from .spec import UNET_DIM_MULTS  # noqa: F401
from .unet import TemporalUnetInference  # noqa: F401
