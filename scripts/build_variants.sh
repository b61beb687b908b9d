#!/bin/bash
# Builds tuning variants of libprism_gpu.so that differ only in the streaming GROUP BY
# kernels' compile-time parameters (both translation units of them).  usage: scripts/build_variants.sh "name:-DFLAG=.. -DFLAG=.." ...
# Output: prism_amd/lib/variants/libprism_gpu_<name>.so (load with PRISM_GPU_LIB=...)
set -eu
cd "$(dirname "$0")/../prism_amd/csrc"
make -j8 > /dev/null
mkdir -p ../lib/variants build/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics $flags \
    -c kernels_stream.hip -o build/variants/kernels_stream_$name.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics $flags \
    -c kernels_stream_pred.hip -o build/variants/kernels_stream_pred_$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  objs=$(ls build/*.o | grep -v kernels_stream.o | grep -v kernels_stream_pred.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/variants/libprism_gpu_$name.so $objs build/variants/kernels_stream_$name.o \
    build/variants/kernels_stream_pred_$name.o
  echo built $name
done
