from .._core import ModelMixin
