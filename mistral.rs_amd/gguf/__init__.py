"""Mirror of `mistralrs-quant/src/gguf/` (GgufMatMul, fast_mmvq, fast_mmq, archive reader)."""
from .qtensor import GgmlDType, QTensor  # noqa: F401
