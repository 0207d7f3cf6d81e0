from ._core import ConfigMixin, register_to_config, FrozenDict
