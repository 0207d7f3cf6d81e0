from .._core import (BaseOutput, logging, SAFETENSORS_WEIGHTS_NAME, WEIGHTS_NAME, USE_PEFT_BACKEND, deprecate,
                     is_torch_version, scale_lora_layers, unscale_lora_layers, is_accelerate_available)
