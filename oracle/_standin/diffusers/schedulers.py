"""`diffusers.schedulers` import surface (hallo/animate/face_animate_static.py:46-50)."""
from ._core import DDIMScheduler  # noqa: F401
from ._core import _OtherScheduler as DPMSolverMultistepScheduler  # noqa: F401
from ._core import _OtherScheduler as EulerAncestralDiscreteScheduler  # noqa: F401
from ._core import _OtherScheduler as EulerDiscreteScheduler  # noqa: F401
from ._core import _OtherScheduler as LMSDiscreteScheduler  # noqa: F401
from ._core import _OtherScheduler as PNDMScheduler  # noqa: F401
