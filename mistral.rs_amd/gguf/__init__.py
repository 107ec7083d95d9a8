"""Mirror of `mistralrs-quant/src/gguf/` (GgufMatMul: matmul.py, fast_mmvq, fast_mmq, fast_gemm, archive reader)."""
from .qtensor import GgmlDType, QTensor  # noqa: F401
