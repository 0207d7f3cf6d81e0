"""Stand-in for the `diffusers` import surface of the Hallo reference (oracle test infrastructure)."""
from ._core import (AutoencoderKL, DDIMScheduler, DiffusionPipeline, ModelMixin, ConfigMixin)
from ._core import _OtherScheduler as DPMSolverMultistepScheduler
from ._core import _OtherScheduler as EulerAncestralDiscreteScheduler
from ._core import _OtherScheduler as EulerDiscreteScheduler
from ._core import _OtherScheduler as LMSDiscreteScheduler
from ._core import _OtherScheduler as PNDMScheduler
__version__ = "0.27.2+standin"
