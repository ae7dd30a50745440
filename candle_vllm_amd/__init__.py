"""candle_vllm_amd -- MI355X (gfx950) paged-attention + quantised-matmul decode path behind the
reference's operator boundary.  Importing the package loads the HIP C-ABI library; there is no fallback."""
from . import _lib  # noqa: F401  (raises if the HIP library is missing or incomplete)
from ._lib import lib, LIB_PATH  # noqa: F401
