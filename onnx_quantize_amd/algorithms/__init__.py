from .gptq import GPTQConfig, _gptq_quantize  # noqa: F401
from .hqq import HqqConfig, _hqq_quantize  # noqa: F401
from .rtn import RTNConfig, _quantize_bias, _rtn_quantize  # noqa: F401
