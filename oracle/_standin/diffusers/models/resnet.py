from .._core import ResnetBlock2D, Downsample2D, Upsample2D
