"""How long does it take just to read the source bytes (48 * n) on this GPU at several sizes?  (floor for the tile kernel)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi

lib = gpa.load()
for n in [1048576, 8388608]:
    p = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    c = torch.zeros((n, 9), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    row = {}
    for mode, name in [(0, "strided_dwords"), (1, "float4"), (2, "lds_dma"), (6, "src_only"), (3, "gather_only"), (4, "src+indep_gather"),
                       (5, "src+dep_gather"), (9, "coop: neither"), (10, "coop: source"), (11, "coop: gather"), (12, "coop: both")]:
        ms = C.c_float()
        best = 1e9
        for _ in range(3):
            _capi.check(_capi.load_tune().gp_debug_stream_bench(C.c_void_p(p.data_ptr()), C.c_void_p(c.data_ptr()), n, mode, 50, C.byref(ms)), "bench")
            best = min(best, ms.value)
        row[name] = f"{best*1e3:8.2f} us = {48*n/best/1e6:7.1f} GB/s"
    print(n, flush=True)
    for k, v in row.items():
        print(f"   {k:22s} {v}", flush=True)
