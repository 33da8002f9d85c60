#!/bin/bash
# build_variant.sh NAME SRC "EXTRA FLAGS": libmmhip with one source rebuilt under extra flags -> benchmarks/bin/libmmhip_NAME.so (A/B runs via MM_LIB)
set -e
cd "$(dirname "$0")/../multimodal_amd/csrc"
mkdir -p ../../benchmarks/bin /tmp/mmvar
extra=""
case "$2" in attention_fast|attention_wide|attention_res) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=off $extra $3 -c $2.hip -o /tmp/mmvar/$2_$1.o
objs=$(ls build/*.o | grep -v "build/$2.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../benchmarks/bin/libmmhip_$1.so $objs /tmp/mmvar/$2_$1.o
echo built benchmarks/bin/libmmhip_$1.so
