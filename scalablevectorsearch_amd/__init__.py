"""scalablevectorsearch_amd -- MI355X-native batched Vamana greedy search behind the SVS operator surface.

Host-side mirror (Python, over the C ABI of include/svsx.h) of the reference interfaces for this one path:
  svs::index::vamana::VamanaIndex::search / VamanaSearchParameters / SearchBufferConfig
  svs::distance::{DistanceL2, DistanceIP, DistanceCosineSimilarity} + distance::compute
  svs::QueryResult, svs::k_recall_at_n
The compute lives in libsvsx.so (hand-written HIP for gfx950); there is no CPU fallback.
"""
from ._lib import SvsxError, build, lib  # noqa: F401
from .distance import DistanceCosineSimilarity, DistanceIP, DistanceL2, DistanceType, compute  # noqa: F401
from .calibrate import CalibrationParameters, calibrate  # noqa: F401
from .iterator import BatchIterator, filtered_search, range_search  # noqa: F401
from .index import (  # noqa: F401
    FlatIndex, QueryResult, SearchBufferConfig, VamanaIndex, VamanaSearchParameters, k_recall_at_n,
)

__all__ = [
    "SvsxError", "build", "lib", "DistanceType", "DistanceL2", "DistanceIP", "DistanceCosineSimilarity", "compute",
    "VamanaIndex", "FlatIndex", "VamanaSearchParameters", "SearchBufferConfig", "QueryResult", "k_recall_at_n", "calibrate", "CalibrationParameters",
    "BatchIterator", "filtered_search", "range_search",
]
