r"""Packed-fingerprint data format helpers (reference: bblean/fingerprints.py:46-108).

Only the pieces either side of the hot path live here: the MSB-first bit packing that
defines the on-device row layout, and the synthetic generator the reference's tests and
benchmarks use.  SMILES -> fingerprint conversion needs RDKit and is out of scope.
"""
from __future__ import annotations

import numpy as np
from numpy.typing import DTypeLike, NDArray

__all__ = ["pack_fingerprints", "unpack_fingerprints", "make_fake_fingerprints"]


def pack_fingerprints(a: NDArray[np.uint8]) -> NDArray[np.uint8]:
    r"""0/1 uint8 array -> bit-packed uint8, most significant bit first
    (``np.packbits``; reference fingerprints.py:46-49)."""
    return np.packbits(a, axis=-1)


def unpack_fingerprints(
    a: NDArray[np.uint8], n_features: int | None = None
) -> NDArray[np.uint8]:
    r"""Bit-packed uint8 -> 0/1 uint8 (reference fingerprints.py:52-67)."""
    return np.unpackbits(a, axis=-1, count=n_features)


def make_fake_fingerprints(
    num: int,
    n_features: int = 2048,
    pack: bool = True,
    seed: int | None = None,
    dtype: DTypeLike = np.uint8,
) -> NDArray[np.uint8]:
    r"""Synthetic fingerprints with a realistic popcount distribution.

    Bit-for-bit the same arrays as the reference generator for the same ``seed``
    (fingerprints.py:70-108) so that its golden vectors can be reused: per-row popcount
    = rint(truncnorm(loc=750, scale=400) clipped to [1, n_features-1]) drawn first,
    then one ``Generator.permuted`` over rows of [1]*pop + [0]*(F-pop).
    """
    import scipy.stats

    if n_features < 1 or n_features % 8 != 0:
        raise ValueError("n_features must be a multiple of 8, and greater than 0")
    if pack and np.dtype(dtype) != np.dtype(np.uint8):
        raise ValueError("Only np.uint8 dtype is supported for packed input")
    loc, scale = 750, 400
    lo, hi = 1, n_features - 1
    rng = np.random.default_rng(seed)
    pops = scipy.stats.truncnorm.rvs(
        (lo - loc) / scale, (hi - loc) / scale, loc=loc, scale=scale, size=num, random_state=rng
    )
    pops = np.rint(pops).astype(np.int64)
    runs = np.empty(2 * num, dtype=np.int64)
    runs[0::2] = pops
    runs[1::2] = n_features - pops
    ordered = np.repeat(np.tile(np.array([1, 0], np.uint8), num), runs)
    fps = rng.permuted(ordered.reshape(num, n_features), axis=-1)
    if pack:
        return np.packbits(fps, axis=1)
    return fps.astype(dtype, copy=False)
