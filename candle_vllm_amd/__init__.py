"""candle_vllm_amd -- MI355X (gfx950) paged-attention + quantised-matmul decode path behind the
reference's operator boundary.  Importing the package loads the HIP C-ABI library; there is no fallback."""
from . import _lib  # noqa: F401  (raises if the HIP library is missing or incomplete)
from ._lib import lib, LIB_PATH  # noqa: F401

import contextlib as _contextlib


class TuningKeyUnavailable(RuntimeError):
    """the key belongs to probe builds (tools/build_probe_lib.sh); the product library honours the ten keys of include/mi355_vllm.h"""


@_contextlib.contextmanager
def tuning(key, value):
    """scoped `mi355_set_tuning`: the key holds `value` inside the block and what it held before afterwards -- also when the block
    raises.  The A/B switches are process-global state of the library; tests use this form.  The library itself knows every product
    key's live value (mi355_get_tuning), so nothing about defaults is restated here."""
    if not lib.mi355_tuning_supported(key):
        raise TuningKeyUnavailable(f"tuning key {key} is not honoured by this build of the library")
    prev = lib.mi355_get_tuning(key)
    if prev == -2 ** 31:                                  # a key this build honours but has no live value for: nothing to restore to
        raise TuningKeyUnavailable(f"tuning key {key} has no recorded default in this build: set it explicitly with mi355_set_tuning")
    lib.mi355_set_tuning(key, value)
    try:
        yield
    finally:
        lib.mi355_set_tuning(key, prev)
