from .._core import ModelMixin, AutoencoderKL
