from .._core import LoRACompatibleConv, LoRACompatibleLinear
