from .._core import Attention, FeedForward, GEGLU, AdaLayerNorm, AdaLayerNormZero
