#!/bin/bash
# tools/build_variant.sh NAME "EXTRA_FLAGS" [network]: builds one network's library with extra
# compiler flags into bnn-pynq_amd/build/variants/NAME/ for A/B timing (BNN_MI355X_LIBDIR=... bench.py)
set -e
NAME=$1; FLAGS=$2; NET=${3:-cnvW1A1}
ROOT=$(cd "$(dirname "$0")/.." && pwd)/bnn-pynq_amd
OUT=$ROOT/build/variants/$NAME; mkdir -p $OUT
declare -A ID=([cnvW1A1]=NET_CNVW1A1 [cnvW1A2]=NET_CNVW1A2 [cnvW2A2]=NET_CNVW2A2 [lfcW1A1]=NET_LFCW1A1 [lfcW1A2]=NET_LFCW1A2)
cd $ROOT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 $FLAGS -c csrc/kernels.hip -o $OUT/kernels.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $FLAGS -DBNN_NETWORK=bnn::${ID[$NET]} -c csrc/runtime.hip -o $OUT/runtime.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/python_sw-$NET-mi355x.so $OUT/runtime.o $OUT/kernels.o build/preprocess.o build/resample.o build/packed_params.o build/faults.o build/topology.o -Wl,--version-script=csrc/exports.map
echo built $OUT/python_sw-$NET-mi355x.so
