#!/bin/bash
# Timing attribution of the fused encoder kernels: build one library per ablation mask HERE (no GPU needed):
#   scripts/ablate_encoder.sh build 1 2 4 8 16 ...   -> vit_amd/librvit_abl<mask>.so
# then on the GPU box:
#   scripts/ablate_encoder.sh run 1 2 4 8 16 ...     -> bench_encoder.py once per library
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  make -C vit_amd/csrc -j8 >/dev/null || exit 1
  for m in "$@"; do
    /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -DENC_ABL=$m -c vit_amd/csrc/encoder.hip -o /tmp/enc_abl$m.o || exit 1
    objs=$(ls vit_amd/csrc/build/*.o | grep -v /encoder.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vit_amd/librvit_abl$m.so $objs /tmp/enc_abl$m.o || exit 1
  done
else
  for m in "$@"; do
    echo "== ENC_ABL=$m"
    RVIT_LIB_PATH=vit_amd/librvit_abl$m.so ROWS=0 timeout -k 10 120 python scripts/bench_encoder.py 2>&1 | grep fused || exit 1
  done
fi
