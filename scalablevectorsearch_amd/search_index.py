"""`search_index` for the GPU path -- same eleven positional arguments as the reference's utils/search_index.cpp:89-178.

  python -m scalablevectorsearch_amd.search_index <query type> <data type> <queries.[fbi]vecs> <search window size> \\
         <neighbours> <threads (ignored)> <config dir> <graph dir|file> <data dir|file> <result prefix> <L2|MIP|Cosine>

Reads what VamanaIndex::save wrote (index/vamana/index.h:825-840): `svs_config.toml` (entry point,
index.h:88-97 / the `entry_point` key of the vamana_index_parameters object) and the native v1 blobs of the dataset
and of the graph (core/io/native.h:307-344; a directory is searched for its single *.svs blob), runs the batch search on
the GPU and writes `<prefix>_idx.ivecs` like QueryResult::save_vecs.
"""
from __future__ import annotations

import glob
import os
import sys
import time

import numpy as np

from . import DistanceCosineSimilarity, DistanceIP, DistanceL2, SearchBufferConfig, VamanaIndex, VamanaSearchParameters, io

HELP = __doc__
_NP = {"int8": np.int8, "uint8": np.uint8, "float": np.float32, "float16": np.float16}
_DIST = {"L2": DistanceL2, "MIP": DistanceIP, "Cosine": DistanceCosineSimilarity}


def _blob(path: str) -> str:
    if os.path.isdir(path):
        files = sorted(glob.glob(os.path.join(path, "*.svs")))
        if len(files) != 1:
            raise SystemExit(f"{path}: expected exactly one .svs blob, found {len(files)}")
        return files[0]
    return path


def _entry_point(config_dir: str) -> int:
    return io.read_index_config(config_dir)["entry_point"]


def load_index(config: str, graph_path: str, data_path: str, distance, data_dtype=np.float32, device: int = 0) -> VamanaIndex:
    """auto_assemble (index/vamana/index.h:1017-1046) for what VamanaIndex::save wrote: svs_config.toml + the native blobs of graph and dataset.
    Like the reference's loader it restores the saved default search parameters (index.h:141-170,318) next to the entry point."""
    cfg = io.read_index_config(config)
    data = io.read_svs_native(_blob(data_path), data_dtype)
    graph = io.read_svs_native(_blob(graph_path), np.uint32)
    index = VamanaIndex(graph, data, distance, [cfg["entry_point"]], device=device)
    sp = cfg["search"]
    if sp is not None:
        index.set_search_parameters(VamanaSearchParameters(SearchBufferConfig(sp["search_window_size"], sp["search_buffer_capacity"]), sp["search_buffer_visited_set"],
                                                           sp["prefetch_lookahead"], sp["prefetch_step"]))
    return index


def _load_queries(path: str, dtype):
    if path.endswith(".fvecs"):
        q = io.read_fvecs(path)
    elif path.endswith(".bvecs"):
        q = io.read_bvecs(path)
    elif path.endswith(".ivecs"):
        q = io.read_ivecs(path)
    else:
        raise SystemExit(f"unsupported query file {path}")
    return q.astype(dtype) if q.dtype != np.dtype(dtype) else q


def main(argv) -> int:
    if len(argv) != 12:
        print(f"Expected 11 arguments. Instead, got {len(argv) - 1}. The required positional arguments are given below.\n{HELP}")
        return 1
    qt, dt, qfile, window, k, _threads, config, graph_path, data_path, prefix, dist = argv[1:]
    if qt not in _NP or dt not in _NP or qt == "float16":
        raise SystemExit(f"Unsupported Query and Data type pair: ({qt}, {dt})!")
    if dist not in _DIST:
        raise SystemExit(f"Unsupported distance type. Valid values: L2/MIP/Cosine. Received: {dist}!")
    data = io.read_svs_native(_blob(data_path), _NP[dt])
    graph = io.read_svs_native(_blob(graph_path), np.uint32)
    index = VamanaIndex(graph, data, _DIST[dist](), [_entry_point(config)])
    index.set_search_parameters(VamanaSearchParameters(SearchBufferConfig(int(window))))
    queries = _load_queries(qfile, _NP[qt])
    print("Running Search.")
    tic = time.perf_counter()
    result = index.search(queries, int(k))
    print(f"Global search time: {time.perf_counter() - tic} seconds")
    ids = result.indices.astype(np.uint32)
    out = np.empty((ids.shape[0], 1 + ids.shape[1]), np.uint32)
    out[:, 0] = ids.shape[1]
    out[:, 1:] = ids
    out.tofile(f"{prefix}_idx.ivecs")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
