"""MI355X-native bundle adjustment behind TheiaSfM's BundleAdjuster API.

The product is the C-ABI shared library built from ``theiasfm_amd/csrc``
(``include/theia_mi355_ba.h``) and the C++ host shim under ``include/theia``.
This Python package is plumbing for tests and the benchmark: the ctypes mirror
of the ABI (``abi``), the library loader (``lib``), synthetic problems
(``synth``), file readers (``io``) and the torch.distributed all-reduce hook
(``dist``).  Nothing here computes bundle adjustment on the CPU.
"""
from . import abi  # noqa: F401
