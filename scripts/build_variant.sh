#!/bin/bash
# Experiment build: vit_amd/librvit_<tag>.so = librvit.so with ONE source recompiled with extra defines
# (usage: scripts/build_variant.sh conv1 v3 -DC1_VARIANT=3).  Never shipped, never loaded by the package itself;
# scripts/bench_*.py load it when RVIT_LIB names it.
set -e
cd "$(dirname "$0")/../vit_amd/csrc"
src=$1; tag=$2; shift 2
make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function "$@" -c $src.hip -o build/${src}_${tag}.var.o
objs=$(ls build/*.o | grep -v "_stamps.o" | grep -v "\.var\.o" | grep -v "build/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librvit_${tag}.so $objs build/${src}_${tag}.var.o
echo built ../librvit_${tag}.so
