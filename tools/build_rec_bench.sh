#!/bin/bash
# builds tools/rec_bench (against the product objects: true times) and tools/rec_bench_prof (against a -DREC_PROF build of
# the kernels, objects under /tmp: where a step's time goes; the clock reads themselves cost ~0.05-0.1 us each)
set -e
cd "$(dirname "$0")/../ocr_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -I ."
make -s crnn_kernels.o
hipcc $F -c ../../tools/rec_bench.hip -o /tmp/rec_bench.o
hipcc --offload-arch=gfx950 -o ../../tools/rec_bench /tmp/rec_bench.o crnn_kernels.o crnn_kernels_bf16.o crnn_kernels_nopk.o
if [ -z "$REC_NOPROF" ]; then
hipcc $F -DREC_PROF $REC_DEFS -c crnn_kernels.hip -o /tmp/crnn_kernels_prof.o
hipcc --offload-arch=gfx950 -o ../../tools/rec_bench_prof /tmp/rec_bench.o /tmp/crnn_kernels_prof.o crnn_kernels_bf16.o crnn_kernels_nopk.o
fi
