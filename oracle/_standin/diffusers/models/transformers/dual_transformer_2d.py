from ..._core import DualTransformer2DModel
