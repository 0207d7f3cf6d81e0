from ._core import VaeImageProcessor
