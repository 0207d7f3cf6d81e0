from .._core import (Attention, AttnProcessor, AttnProcessor2_0, AttentionProcessor, ADDED_KV_ATTENTION_PROCESSORS, CROSS_ATTENTION_PROCESSORS, AttnAddedKVProcessor)
