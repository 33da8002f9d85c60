#!/bin/bash
# build_variant.sh <tag> [extra hipcc flags...]: an A/B build of libmmhip.so with extra -D flags ->
# multimodal_amd/lib/libmmhip_<tag>.so (select it at run time with MM_LIB=<path>).
set -e
tag=$1; shift
cd "$(dirname "$0")/../multimodal_amd/csrc"
mkdir -p build_$tag ../lib
for f in *.hip; do
  o=build_$tag/${f%.hip}.o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$REBUILD" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=off "$@" -c $f -o $o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmmhip_$tag.so build_$tag/*.o
echo built ../lib/libmmhip_$tag.so
