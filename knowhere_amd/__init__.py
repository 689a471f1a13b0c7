"""knowhere_amd -- MI355X-native ANN Search() backend for Knowhere.

The product is ``libknhip.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/knhip.h``) and the C++ IndexNode in ``knowhere_amd/host``.  This Python package is the
thin harness the tests and ``bench.py`` drive it through: ctypes bindings, torch for device
memory / streams / ``torch.distributed`` plumbing, and the GPU index builder.
"""
from . import _lib  # noqa: F401
from .index import GpuIndex, KnhipError, RowStore  # noqa: F401
