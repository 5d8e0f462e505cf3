#!/bin/bash
# Build a variant of libuic_hip.so for A/B runs in one process-per-variant sweep:
#   tools/build_variant.sh <name> <file.hip> "<extra hipcc flags>"   ->   variants/libuic_<name>.so   (run with UIC_LIB=...)
# Only <file.hip> is recompiled (with the extra flags); every other object comes from the last in-tree build.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; flags=$3
mkdir -p variants
C=unpaired_image_captioning_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $flags -c $C/$src -o variants/${src%.hip}_$name.o
objs=""
for o in $C/*.o; do
  if [ "$(basename $o)" = "${src%.hip}.o" ]; then objs="$objs variants/${src%.hip}_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libuic_$name.so $objs -lz -lpthread -ldl
rm -f variants/${src%.hip}_$name.o
echo variants/libuic_$name.so
