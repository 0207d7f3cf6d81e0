from .._core import get_activation
