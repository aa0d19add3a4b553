"""gemma_amd -- MI355X-native (gfx950) kinship + univariate-LMM path of GEMMA.

The product is gemma_amd/libgemma_hip.so (hand-written HIP; C ABI in include/gemma_hip.h).  This
package is the thin host-side mirror of the reference interfaces on that path (gemma_amd.api),
the build recipe (gemma_amd.build) and SNP-shard helpers for one-process-per-GPU runs
(gemma_amd.dist).  Importing the package does not require a GPU; calling into it does.
"""
from . import _lib  # noqa: F401
from ._lib import GemmaHipError  # noqa: F401

__all__ = ["api", "build", "dist", "GemmaHipError"]
