"""What one reported byte of rocprofv3's FETCH_SIZE stands for, by access kind, as calibrated on the MI355X box with
profiles/experiments/fetch_calibration.hip (profiles/fetch_calibration.sh -> profiles/r06_fetch_calibration.json; every
kernel reads a known byte count once from arrays of 0.5 - 3 GB, far beyond the 256 MiB Infinity Cache):

    consecutive lanes, consecutive addresses, 16 B or 4 B per lane      reported = 0.500 of the bytes read   -> x 2
    a 32-byte record per lane at a scattered index (2 x 16 B)           reported = 2.05 x the bytes read: a 64-byte line per record
    two 32-byte records per lane from two arrays (the blends' gathers)  reported = 2.00 x: two lines per entry
    a 64-byte aligned record per lane                                   reported = 1.00 x
    a 192-byte row per lane (12 x 16 B, the SH rows)                    reported = 1.007 x

i.e. FETCH_SIZE is the count of memory-side read requests x 64 B: exact for scattered line-sized requests, half for the
128-byte requests a wide coalesced stream is served with.  The HBM bytes of a kernel are therefore 2 x FETCH_SIZE for a
stream and 1 x FETCH_SIZE for a gather kernel (whose requested bytes are the LINES, not the 32 bytes used of each).
KIND names the dominant read pattern of every kernel; "mixed" kernels get the stream factor (an upper bound) and say so."""

FACTOR = {"stream": 2.0, "gather": 1.0, "mixed": 2.0}
KIND = {
    # geometry: per-Gaussian inputs read by consecutive lanes
    "k_preprocess_fwd": "stream", "k_preprocess_bwd": "stream", "k_preprocess_bwd_common": "stream", "k_offset_reduce": "stream",
    "k_grads_rezero": "stream",
    # SH rows / accumulator rows / records of scattered Gaussians
    "k_appearance": "gather", "k_preprocess_bwd_rows": "gather",
    "k_render_fwd": "gather", "k_render_fwd_seg": "gather", "k_render_bwd": "gather", "k_tail_build": "gather",
    # binning: rectangles and depth bits by consecutive lanes
    "k_super_bin<0>": "stream", "k_super_bin<1>": "stream", "k_super_bin<2>": "stream", "k_tile_count": "stream", "k_tile_scatter": "stream",
    # entry lists read as streams, depth bits of the entries' Gaussians gathered
    "k_tile_pull": "mixed", "k_tile_front": "mixed", "k_tile_tail": "mixed", "k_tile_sort_small": "stream", "k_tile_sort_big": "stream",
    "k_tile_order": "stream",
}


def kind_of(kernel):
    return KIND.get(kernel, "stream")


def hbm_bytes(kernel, fetch_raw, write):
    """(HBM bytes, kind, factor): factor x FETCH_SIZE + WRITE_SIZE (WRITE_SIZE reads the bytes exactly: MI355X_MICROARCH.md)"""
    k = kind_of(kernel)
    return FACTOR[k] * fetch_raw + write, k, FACTOR[k]
