r"""Merge criteria of the BitBIRCH leaf test (reference: bblean/_merges.py).

In the reference each criterion is a Python callable evaluated on the host for every
leaf-merge attempt.  Here the decision runs inside the device insertion kernel
(csrc/bb_tree.hip, `merge_accept`), so on the host a criterion is only a *descriptor*:
an integer code, the tolerance, and - for the adaptive-tolerance criteria - the table
``tol[old_n] = max(tolerance * (exp(-decay*old_n) - exp(-decay*n_max)), 0)`` evaluated
with NumPy exactly as `_merges.py:82-116` does, so the device never calls ``exp``.
"""
from __future__ import annotations

import dataclasses
import functools

import numpy as np
from numpy.typing import NDArray

BUILTIN_MERGES = [
    "radius",
    "diameter",
    "tolerance-diameter",
    "tolerance-radius",
    "tolerance-legacy",
    "never-merge",
]

# codes shared with include/bbhip.h (BBH_CRIT_*) and oracle/bb_oracle.h (BBO_CRIT_*)
CRITERION_CODES = {
    "diameter": 0,
    "radius": 1,
    "tolerance-diameter": 2,
    "tolerance-radius": 3,
    "tolerance-legacy": 4,
    "never-merge": 5,
}
_HAS_TOLERANCE = {"tolerance-diameter", "tolerance-radius", "tolerance-legacy", "never-merge"}
_ADAPTIVE = {"tolerance-diameter", "tolerance-radius", "never-merge"}


@dataclasses.dataclass
class MergeCriterion:
    r"""Descriptor of a merge criterion (stands in for `MergeAcceptFunction`,
    _merges.py:19-37)."""

    name: str
    _tolerance: float | None = None
    n_max: int = 1000
    decay: float = 1e-3

    @property
    def code(self) -> int:
        return CRITERION_CODES[self.name]

    @property
    def has_tolerance(self) -> bool:
        return self.name in _HAS_TOLERANCE

    @property
    def tolerance(self) -> float | None:
        return self._tolerance if self.has_tolerance else None

    @tolerance.setter
    def tolerance(self, value: float) -> None:
        self._tolerance = value

    def tolerance_table(self) -> NDArray[np.float64]:
        r"""tol(old_n) for old_n in [0, n_max]; 0 beyond (exp is decreasing, so the
        reference's max(..., 0.0) clamps every later entry, _merges.py:113)."""
        if self.name not in _ADAPTIVE or self._tolerance is None:
            return np.zeros(0, dtype=np.float64)
        return _tolerance_table(float(self._tolerance), float(self.decay), int(self.n_max))

    def __repr__(self) -> str:
        cls = {
            "radius": "RadiusMerge",
            "diameter": "DiameterMerge",
            "tolerance-diameter": "ToleranceDiameterMerge",
            "tolerance-radius": "ToleranceRadiusMerge",
            "tolerance-legacy": "ToleranceMerge",
            "never-merge": "NeverMerge",
        }[self.name]
        if self.name in ("tolerance-diameter", "tolerance-radius", "tolerance-legacy"):
            return f"{cls}({self._tolerance})"
        return f"{cls}()"


@functools.lru_cache(maxsize=64)
def _tolerance_table(tolerance: float, decay: float, n_max: int) -> NDArray[np.float64]:
    # scalar np.exp per entry, as the reference evaluates it (a vectorised exp may round differently); every tree of
    # a multiround run asks for the same table, hence the cache
    offset = np.exp(-decay * n_max)
    out = np.empty(n_max + 1, dtype=np.float64)
    for old_n in range(n_max + 1):
        out[old_n] = max(tolerance * (np.exp(-decay * old_n) - offset), 0.0)
    out.setflags(write=False)
    return out


def get_merge_accept_fn(merge_criterion: str, tolerance: float = 0.05) -> MergeCriterion:
    r"""Factory with the reference's name and error (``_merges.py:194-212``)."""
    if merge_criterion not in CRITERION_CODES:
        raise ValueError(
            f"Unknown merge criterion {merge_criterion} "
            "Valid criteria are: radius|diameter|tolerance-diameter|tolerance-radius"
        )
    if merge_criterion in _HAS_TOLERANCE:
        return MergeCriterion(merge_criterion, tolerance)
    return MergeCriterion(merge_criterion, None)
