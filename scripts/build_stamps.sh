#!/bin/bash
# Diagnosis build: vit_amd/librvit_stamps.so = librvit.so with ONE source recompiled with a stamp define
# (usage: scripts/build_stamps.sh conv3x3_p8 RVIT_C8_STAMPS).  Never shipped, never loaded by the package itself.
set -e
cd "$(dirname "$0")/../vit_amd/csrc"
src=${1:-conv3x3_p8}; def=${2:-RVIT_C8_STAMPS}
make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function -D$def -c $src.hip -o build/${src}_stamps.o
objs=$(ls build/*.o | grep -v "_stamps.o" | grep -v "build/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librvit_stamps.so $objs build/${src}_stamps.o
echo built ../librvit_stamps.so
