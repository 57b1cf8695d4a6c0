#!/usr/bin/env python3
"""Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for the node pool's access shape (run under
`rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 tools/calibrate_pmc.py`).
Prints the exactly known byte and cache-line counts of the probe so the counter values can be set against them."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthesis_amd as sa  # noqa: E402
from synthesis_amd.engine import load_library  # noqa: E402
import ctypes as C  # noqa: E402

n_records = (6 << 30) // 32           # 6 GiB of 32-byte records: far beyond L2 (32 MiB) and Infinity Cache (256 MiB)
n_spans = 8_000_000
rng = np.random.RandomState(0)
off = rng.randint(0, n_records - 16, size=n_spans).astype(np.uint32)
buf = torch.zeros(n_records * 8, dtype=torch.float32, device="cuda")
d_off = torch.from_numpy(off.view(np.int32)).cuda()
eng = sa.Engine(concurrent_games=16, max_explores=8)
lib = load_library()
first = off.astype(np.int64) * 32
last = first + 9 * 32 - 1
out = {"n_spans": n_spans, "useful_bytes": n_spans * 288,
       "lines64_bytes": int(((last // 64) - (first // 64) + 1).sum() * 64),
       "lines128_bytes": int(((last // 128) - (first // 128) + 1).sum() * 128)}
for do_write in (0, 1):
    torch.cuda.synchronize()
    rc = lib.syn_debug_calibrate(eng._h, C.c_void_p(buf.data_ptr()), C.c_void_p(d_off.data_ptr()), n_spans, do_write)
    assert rc == 0
    out[f"kernel_ms_write{do_write}"] = eng.last_kernel_ms()
out["written_bytes_when_write1"] = n_spans * 9 * 16
print("CALIB " + json.dumps(out))
