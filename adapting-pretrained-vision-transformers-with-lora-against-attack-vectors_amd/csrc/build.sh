#!/bin/bash
# Build libvitlora_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libvitlora_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
mkdir -p build
pids=()
for f in gemm gemm256 gemm_pp gemm_stream elementwise attention32 cls_path lora_grad f32_kernels patch swin vitlora_f32 vitlora; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ kernels.h -nt build/$f.o ] || [ gemm.h -nt build/$f.o ] || [ prof.h -nt build/$f.o ] || [ model.h -nt build/$f.o ] || [ f32_kernels.h -nt build/$f.o ] || [ gemm_epi.h -nt build/$f.o ] || [ ../../include/vitlora.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT build/gemm.o build/gemm256.o build/gemm_pp.o build/gemm_stream.o build/elementwise.o build/attention32.o build/cls_path.o build/lora_grad.o build/f32_kernels.o build/patch.o build/swin.o build/vitlora_f32.o build/vitlora.o
echo "built $OUT"
