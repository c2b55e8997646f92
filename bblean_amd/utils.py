r"""Small host helpers mirrored from the reference (bblean/utils.py)."""
from __future__ import annotations

import itertools
import typing as tp

import numpy as np

__all__ = ["min_safe_uint", "batched", "hip_extension_is_available"]

_T = tp.TypeVar("_T")


def min_safe_uint(nmax: int) -> np.dtype:
    r"""Smallest unsigned dtype holding ``nmax`` (reference: bblean/utils.py:25-34).

    The BitFeature buffers that cross the API (``_bf_to_np``, multiround ``round-*``
    files) use exactly this dtype for a cluster of ``nmax`` samples.
    """
    dt = np.min_scalar_type(nmax)
    if dt.hasobject:
        raise ValueError(f"n_samples: {nmax} is too large to hold in a uint64 array")
    return dt


def batched(iterable: tp.Iterable[_T], n: int) -> tp.Iterator[tuple[_T, ...]]:
    r"""itertools.batched for Python < 3.12 (reference: bblean/utils.py:38-48)."""
    if n < 1:
        raise ValueError("n must be at least one")
    it = iter(iterable)
    while chunk := tuple(itertools.islice(it, n)):
        yield chunk


def hip_extension_is_available() -> bool:
    r"""Whether libbbhip.so is built and loadable (analogue of
    ``cpp_extensions_are_installed``, bblean/utils.py:123-130)."""
    try:
        from bblean_amd import _lib

        _lib.load()
        return True
    except Exception:
        return False


def release_device_cache() -> None:
    r"""Return the device blocks libbbhip keeps for reuse to the driver (``bbh_trim_cache``)."""
    from bblean_amd import _lib

    _lib.check(_lib.load().bbh_trim_cache())
