"""sc_table_upload / sc_table_download rates (the boundary's host-buffer calls)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from __graft_entry__ import load_package

pkg = load_package()
ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
for n in (20, 24, 27, 28):
    host = np.arange(1 << n, dtype=np.uint64)
    for rep in range(3):
        t0 = time.perf_counter()
        t = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, host)
        up = time.perf_counter() - t0
        t0 = time.perf_counter()
        back = t.to_evaluations()
        down = time.perf_counter() - t0
        assert back[12345 % back.size] == host[12345 % host.size] and back[-1] == host[-1]
        del t
    print("n=%d  %7.1f MiB  upload %8.2f ms = %6.2f GB/s   download %8.2f ms = %6.2f GB/s" % (
        n, host.nbytes / 2**20, up * 1e3, host.nbytes / up / 1e9, down * 1e3, host.nbytes / down / 1e9), flush=True)
