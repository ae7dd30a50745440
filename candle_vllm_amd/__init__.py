"""candle_vllm_amd -- MI355X (gfx950) paged-attention + quantised-matmul decode path behind the
reference's operator boundary.  Importing the package loads the HIP C-ABI library; there is no fallback."""
from . import _lib  # noqa: F401  (raises if the HIP library is missing or incomplete)
from ._lib import lib, LIB_PATH  # noqa: F401

import contextlib as _contextlib

# defaults of the tuning keys whose default is not 0 (include/mi355_vllm.h): what `tuning` restores when a key was never set before
_TUNING_DEFAULTS = {3: 1, 6: 1, 9: 1, 10: 1024, 11: 2, 12: 96, 14: 1, 17: 2, 21: 8, 36: 96, 38: 4, 41: 1, 42: 1, 44: 1, 47: 1, 48: 1}


@_contextlib.contextmanager
def tuning(key, value):
    """scoped `mi355_set_tuning`: the key holds `value` inside the block and what it held before (or its default) afterwards --
    also when the block raises.  The A/B switches are process-global state of the library; tests use this form."""
    prev = lib.mi355_get_tuning(key)
    if prev == -2 ** 31:
        prev = _TUNING_DEFAULTS.get(key, 0)
    lib.mi355_set_tuning(key, value)
    try:
        yield
    finally:
        lib.mi355_set_tuning(key, prev)
