from .._core import AdaLayerNormSingle, AdaLayerNorm, AdaLayerNormZero
