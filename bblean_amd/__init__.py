r"""bblean_amd - MI355X-native BitBIRCH similarity / insertion engine.

Drop-in for the hot path of mqcomplab/bblean: `bblean_amd.BitBirch` mirrors
`bblean.bitbirch.BitBirch`, `bblean_amd.similarity` mirrors `bblean.similarity`; both
run on hand-written HIP kernels (csrc/) through the C ABI in include/bbhip.h.
"""
from bblean_amd.bitbirch import BitBirch, fit_buffers_concurrently, fit_concurrently, set_merge
from bblean_amd.fingerprints import (
    make_fake_fingerprints,
    pack_fingerprints,
    unpack_fingerprints,
)

__version__ = "0.1.0"

__all__ = [
    "BitBirch",
    "set_merge",
    "fit_concurrently",
    "fit_buffers_concurrently",
    "pack_fingerprints",
    "unpack_fingerprints",
    "make_fake_fingerprints",
    "__version__",
]
