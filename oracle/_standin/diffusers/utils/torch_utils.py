from .._core import randn_tensor, apply_freeu
