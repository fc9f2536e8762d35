#!/usr/bin/env python3
"""FETCH_SIZE calibration on known byte counts (MI355X_MICROARCH.md: "calibrate on a
known byte count in your own access pattern before trusting an absolute").

Run under `rocprofv3 --pmc FETCH_SIZE -- python3 tools/pmc_calib.py`: builds a
small index (the data is irrelevant), launches k_hbm_read (16 B per lane) and
k_hbm_read_x2 (8 B per lane = global_load_dwordx2, the scan kernels' width) once
over the same number of bytes, far beyond the Infinity Cache, and prints it.
tools/pmc_summary.py --calib turns the two counter values into the factors."""
import ctypes as C
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nxsearch_amd as N                      # noqa: E402
from nxsearch_amd import corpus               # noqa: E402

work = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir(),
                    "nxs_calib_%d" % os.getuid())
info = corpus.write_corpus(work, 4_000_000, 200_000, seed=0)
nxs = N.Nxs(work)
idx = nxs.open_files(info["terms"], info["dtmap"])
L = N.lib()
L.nxsgpu_hbm_calibrate.restype = C.c_int
L.nxsgpu_hbm_calibrate.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
b = C.c_uint64()
rc = L.nxsgpu_hbm_calibrate(idx.device, C.byref(b))
print(json.dumps({"rc": rc, "bytes_per_kernel": b.value, "postings": info["postings"]}))
idx.close()
nxs.close()
shutil.rmtree(work, ignore_errors=True)
