#!/bin/bash
# Timing ablations of the deformation network's fp16 forward: builds libgftorf_rast.so variants with -DDF_ABL=<bits> into
# gftorf_amd/_abl/ (here, no GPU needed) ...      bash profiles/deform_ablate.sh build
# ... and times each on the GPU box ...          bash profiles/deform_ablate.sh run > gpurun_out/deform_ablate.txt
# bits: 1 = minimal epilogue (no bias / ReLU / split / saves), 2 = every weight load from the segment's first two chunks
# (L1 hits instead of L2), 4 = no saved activations, 8 = one multiply per tile instead of three.  Results are wrong by design.
set -u
cd "$(dirname "$0")/.."
VARS="${VARS:-0 1 2 3 4 8 10 11 16}"
if [ "${1:-}" = build ]; then
    mkdir -p gftorf_amd/_abl
    # the variants are built over the product library: whatever happens, it comes back (a failed or interrupted build
    # must not leave a wrong-by-design library installed) and the objects are rebuilt from the unflagged sources
    cp gftorf_amd/libgftorf_rast.so /tmp/libgft_keep.so
    trap 'cp /tmp/libgft_keep.so gftorf_amd/libgftorf_rast.so; touch gftorf_amd/csrc/k_deform.hip' EXIT INT TERM
    for v in $VARS; do
        touch gftorf_amd/csrc/k_deform.hip
        GFT_EXTRA_FLAGS="-DDF_ABL=$v" python3 -c "from gftorf_amd import build; build.build()" && cp gftorf_amd/libgftorf_rast.so gftorf_amd/_abl/lib_$v.so && echo built $v
    done
    touch gftorf_amd/csrc/k_deform.hip
    python3 -c "from gftorf_amd import build; build.build()"
else
    for v in $VARS; do
        echo "DF_ABL=$v"
        GFT_ABL_LIB=gftorf_amd/_abl/lib_$v.so timeout -k 10 120 python3 profiles/deform_ab.py 300000 noerr
    done
fi
