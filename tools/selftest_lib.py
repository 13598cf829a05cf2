"""(GPU box) run pbf_selftest_math of the library named by PBF_HIP_LIB (or the default) and print the mismatch counts."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

import bench  # noqa: E402

pkg = bench.load_package()
s = pkg.Solver(h=0.1)
bad = np.zeros(4, np.uint64)
s._chk(s.L.pbf_selftest_math(s.ctx, bad.ctypes.data_as(C.c_void_p)), "pbf_selftest_math")
print("mismatches [sqrt, (h-r)^2/r, x/poly6(0.3h), x/RHO]:", bad.tolist())
