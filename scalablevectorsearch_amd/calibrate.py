"""Search-parameter calibration: mirrors index/vamana/calibrate.h for the buffer parameters.

  optimize_search_buffer                      calibrate.h:249-343   lower_bound over the window for the target recall
  optimize_split_buffer_using_binary_search   calibrate.h:150-169   window < k with capacity fixed to k
Prefetch tuning (calibrate.h:345-470) has no GPU analogue (latency is hidden by wave occupancy) and is omitted.
Host logic only: `compute_recall(params) -> float` runs the search (on the GPU) and scores it.
"""
from __future__ import annotations

from dataclasses import dataclass, replace

from .index import SearchBufferConfig, VamanaSearchParameters


@dataclass
class CalibrationParameters:
    """calibrate.h:91-109 (buffer-related fields)."""

    search_window_size_upper: int = 1000
    search_window_capacity_upper: int = 1000
    search_buffer_optimization: str = "All"  # Disable | All | ROIOnly | ROITuneUp
    use_existing_parameter_values: bool = True


def _lower_bound(lo: int, hi: int, pred_is_less):
    """std::lower_bound over the integer range [lo, hi): first v for which pred_is_less(v) is False (hi if none)."""
    while lo < hi:
        mid = lo + (hi - lo) // 2
        if pred_is_less(mid):
            lo = mid + 1
        else:
            hi = mid
    return lo


def optimize_search_buffer(cp: CalibrationParameters, current: VamanaSearchParameters, num_neighbors: int, target_recall: float, compute_recall):
    """Returns (parameters, converged).  calibrate.h:249-343 for datasets without reranking."""
    state = cp.search_buffer_optimization
    cur_cap = current.buffer_config.total_capacity
    use_current_capacity = state == "ROITuneUp"
    max_recall = float("-inf")

    def configured(window):
        cap = max(window, cur_cap) if use_current_capacity else window
        return replace(current, buffer_config=SearchBufferConfig(window, cap))

    lower = 1 if (use_current_capacity and cur_cap >= num_neighbors) else num_neighbors

    def is_less(window):
        nonlocal max_recall
        r = compute_recall(configured(window))
        max_recall = max(max_recall, r)
        return r < target_recall

    upper = cp.search_window_size_upper
    window = _lower_bound(lower, upper, is_less)
    window = min(window, upper - 1) if upper > lower else lower  # *lower_bound(...) of a UnitRange never dereferences end
    converged = max_recall >= target_recall
    best = configured(window)
    if state != "All" or not converged or window != num_neighbors:
        return best, converged
    # "maybe oversized": the window already equals k -> shrink the window below k with capacity fixed to k
    cap = best.buffer_config.total_capacity
    w = _lower_bound(1, cap, lambda ws: compute_recall(replace(best, buffer_config=SearchBufferConfig(ws, cap))) < target_recall)
    w = min(w, cap - 1) if cap > 1 else 1
    return replace(best, buffer_config=SearchBufferConfig(w, cap)), converged


def calibrate(index, queries, groundtruth, num_neighbors: int, target_recall: float, cp: CalibrationParameters | None = None):
    """VamanaIndex::calibrate analogue: tunes index's default search parameters for `target_recall` on host arrays."""
    from .index import k_recall_at_n

    cp = cp or CalibrationParameters()

    def compute_recall(p):
        return k_recall_at_n(groundtruth, index.search(queries, num_neighbors, p).indices, num_neighbors, num_neighbors)

    start = index.get_search_parameters() if cp.use_existing_parameter_values else VamanaSearchParameters()
    if cp.search_buffer_optimization == "Disable":
        return start
    best, converged = optimize_search_buffer(cp, start, num_neighbors, target_recall, compute_recall)
    index.set_search_parameters(best)
    return best
