"""nirgan_hip: host side of the MI355X-native NIR-GAN Pix2Pix path.

Imports libnirgan_hip.so through ctypes (``lib``); fails loudly when the library has not been
built.  ``engine``/``nets`` build launch plans, ``trainer`` runs the fused two-optimizer step,
``parallel`` shards tile batches over ranks with RCCL all-reduce of flat gradients.
"""
from . import lib  # noqa: F401  (raises ImportError when the HIP library is missing)

__all__ = ["lib"]
