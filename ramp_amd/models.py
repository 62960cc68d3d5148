"""Namespace mirroring ``mpd.models`` for the sampler hot path."""
from .diffusion import GaussianDiffusionModel3d, StaticGaussianDiffusionModel  # noqa: F401
from .spec import UNET_DIM_MULTS  # noqa: F401
from .unet import TemporalUnetInference  # noqa: F401
