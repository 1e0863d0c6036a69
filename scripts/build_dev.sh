#!/bin/bash
# Dev build of libqspec_hip.so with extra -D flags (in-kernel stamps etc.) into build_dev/ (git-ignored).
#   scripts/build_dev.sh stamps "-DQS_SMH_STAMPS"   ->  build_dev/libqspec_hip_stamps.so   (use via QSPEC_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
name=${1:-dev}; extra=${2:-}
mkdir -p build_dev/$name
for f in capi norm_quant hadamard gemm gemm_stream gemm_tiled attention sampler comm; do
  src=qspec_amd/csrc/$f.hip; obj=build_dev/$name/$f.o
  if [ ! -f $obj ] || [ $src -nt $obj ] || [ qspec_amd/csrc/common.cuh -nt $obj ] || [ "$FORCE" = 1 ]; then
    /opt/rocm/bin/hipcc $extra -DQS_EXPERIMENTAL -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math \
      -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Iqspec_amd/csrc -c $src -o $obj &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_dev/libqspec_hip_$name.so build_dev/$name/*.o
echo built build_dev/libqspec_hip_$name.so
