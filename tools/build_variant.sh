#!/bin/bash
# A variant of the library whose tile kernels (cs_corr_mfma.hip) are compiled with extra flags:
#   bash tools/build_variant.sh <name> "<flags>"   -> chromosight_amd/csrc/build/libchromosight_hip_<name>.so
name=$1; flags=$2
cd "$(dirname "$0")/../chromosight_amd/csrc"
make -s -j8 || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-value -DCS_HAVE_FAST $flags -c cs_corr_mfma.hip -o build/cs_corr_mfma_$name.o || exit 1
objs=$(ls build/*.o | grep -v "cs_corr_mfma" )
hipcc --offload-arch=gfx950 -shared -fPIC $objs build/cs_corr_mfma_$name.o -ldl -o build/libchromosight_hip_$name.so
ls -la build/libchromosight_hip_$name.so
