from .._core import is_xformers_available, is_accelerate_available
