"""FETCH_SIZE calibration for gfx950 (MI355X_MICROARCH.md 'HBM'): stream a known byte count through the same
row-group access pattern the search kernel uses (G lanes x 16 B per row) and compare with the counter.
Usage (on the GPU box):  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_calib.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scalablevectorsearch_amd as svs  # noqa: E402

nb, d = 3_000_000, 96  # 1.152 GB of rows, read exactly once by distance_kernel
b = np.random.default_rng(0).standard_normal((nb, d)).astype(np.float32)
a = np.ones((1, d), np.float32)
out = svs.compute(svs.DistanceL2(), a, b)
print("calib rows", nb, "bytes", nb * d * 4, "checksum", float(out.sum()))
