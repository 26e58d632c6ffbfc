#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"  -> ab/libhybkf_<name>.so built from the working tree (A/B runs on ONE gpu box:
# boxes differ by up to ~15 %, so two variants are only comparable inside one gpurun call: KF_LIB=$PWD/ab/libhybkf_<name>.so)
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
mkdir -p ab/obj_$name
for f in ctx preprocess track integrate raycast mcubes; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-value -Iinclude $flags -c hybkinectfu_amd/csrc/$f.hip -o ab/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libhybkf_$name.so ab/obj_$name/*.o
echo built ab/libhybkf_$name.so
