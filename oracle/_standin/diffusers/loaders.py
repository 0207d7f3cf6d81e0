from ._core import UNet2DConditionLoadersMixin
