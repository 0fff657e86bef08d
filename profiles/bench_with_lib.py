#!/usr/bin/env python3
"""Runs bench.py on an experiment build of the library:
    python -m gftorf_amd.build --tag noprefetch -DGFT_FWD_PREFETCH=0        (here: cross-compiles)
    GFT_ABL_LIB=gftorf_amd/_abl/lib_noprefetch.so python profiles/bench_with_lib.py --workload fog --steps 50    (GPU box)
Without GFT_ABL_LIB it is bench.py on the product library: two runs in one gpurun call are an A/B on one box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gftorf_amd import _lib  # noqa: E402

if os.environ.get("GFT_ABL_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["GFT_ABL_LIB"])
    print("library:", _lib.LIB_PATH, file=sys.stderr)
    if os.environ.get("GFT_ABL_ANY_ABI"):
        # (an older build whose argument blocks are the same: A/B across an ABI bump that only changed what a buffer holds)
        import ctypes
        _lib.ABI_VERSION = ctypes.CDLL(_lib.LIB_PATH).gft_abi_version()
import bench  # noqa: E402

bench.main()
