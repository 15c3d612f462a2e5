#!/bin/bash
# Library variant for same-box A/B runs: tools/build_variant.sh <name> <source stems, comma separated> "<extra hipcc flags>"
#   compiles csrc/<stem>.hip (both 16-bit builds) with the extra flags and links them with the other, default objects into
#   csrc/build/lib<name>.so; select it with VITLORA_LIB=$PWD/<package>/csrc/build/lib<name>.so (tools/ab_libs.sh).
set -e
P=adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd
cd "$(dirname "$0")/../$P/csrc"
NAME=$1; STEMS=${2//,/ }; EXTRA=$3
bash build.sh > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
for STEM in $STEMS; do
  TR=""
  case "$STEM" in gemm256|elementwise) TR="-mllvm -amdgpu-use-amdgpu-trackers=1";; esac
  TRB="$TR"; [ "$STEM" = gemm256 ] && TRB="$TR -mllvm -amdgpu-schedule-relaxed-occupancy=true"
  hipcc $FLAGS $TR $EXTRA -c $STEM.hip -o build/${STEM}__$NAME.o &
  case "$STEM" in f32_kernels|patch|swin|vitlora_f32) ;; *) hipcc $FLAGS $TRB $EXTRA -DVL_BF16 -c $STEM.hip -o build/${STEM}_bf16__$NAME.o & ;; esac
done
wait
OBJS=""
for o in build/*.o; do
  case "$o" in *__*) continue;; esac
  b=$(basename $o .o)
  if [ -f build/${b}__$NAME.o ]; then OBJS="$OBJS build/${b}__$NAME.o"; else OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o build/lib$NAME.so $OBJS
echo "built $P/csrc/build/lib$NAME.so"
