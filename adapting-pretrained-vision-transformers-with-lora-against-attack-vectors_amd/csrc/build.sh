#!/bin/bash
# Build libvitlora_hip.so for gfx950 (cross-compiles without a GPU).
# The 16-bit operand path is compiled TWICE from the same sources: fp16 operands (namespace vl_f16, also carries the fp32 parity
# mode, Swin-T and the handle-less entry points) and -DVL_BF16 (namespace vl_bf16); api_dispatch.cpp (generated) exports the ABI.
set -e
cd "$(dirname "$0")"
OUT=../libvitlora_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
mkdir -p build
python3 gen_dispatch.py > /dev/null
HDRS="common.h kernels.h gemm.h prof.h model.h f32_kernels.h gemm_epi.h api_rename.h mlp_fused.h ../../include/vitlora.h"
stale() {   # stale <object> <source>
  [ ! -f "$1" ] && return 0
  [ "$2" -nt "$1" ] && return 0
  for h in $HDRS; do [ "$h" -nt "$1" ] && return 0; done
  return 1
}
BOTH="gemm gemm256 gemm_pp gemm_stream elementwise attention32 cls_path lora_grad vitlora"
ONCE="f32_kernels patch swin vitlora_f32 mlp_fused"
pids=()
njobs=0
run() { "$@" & pids+=($!); njobs=$((njobs + 1)); if [ $njobs -ge 8 ]; then wait -n || exit 1; njobs=$((njobs - 1)); fi; }
OBJS=""
# per-file flags.  gemm256.hip sits at exactly 256 VGPRs: with the scheduler's AMDGPU register-pressure trackers the fp16 build's
# GELU-forward instantiation loses its three spilled VGPRs (a scratch reload is an s_waitcnt vmcnt(0) that drains the LDS-DMA queue;
# -6 % on that kernel, same box).  The bf16 build of the same file GAINS ten spills in two other epilogues with that flag alone
# and is spill-free with relaxed-occupancy scheduling added (allocation at the register edge is not monotone: both tables are
# pinned by tests/test_host_cpu.py).  elementwise.hip: the LayerNorm backward schedules 7 % faster with the trackers (both builds).
extra() {   # extra <file> <build: f16 | bf16>
  case "$1:$2" in
    gemm256:f16|elementwise:f16|elementwise:bf16) echo "-mllvm -amdgpu-use-amdgpu-trackers=1";;
    gemm256:bf16) echo "-mllvm -amdgpu-use-amdgpu-trackers=1 -mllvm -amdgpu-schedule-relaxed-occupancy=true";;
    *) echo "";;
  esac
}
for f in $BOTH $ONCE; do
  OBJS="$OBJS build/$f.o"
  if stale build/$f.o $f.hip; then run hipcc $FLAGS $(extra $f f16) -c $f.hip -o build/$f.o; fi
done
for f in $BOTH; do
  OBJS="$OBJS build/${f}_bf16.o"
  if stale build/${f}_bf16.o $f.hip; then run hipcc $FLAGS $(extra $f bf16) -DVL_BF16 -c $f.hip -o build/${f}_bf16.o; fi
done
OBJS="$OBJS build/api_dispatch.o"
if stale build/api_dispatch.o api_dispatch.cpp; then run hipcc -O2 -std=c++17 -fPIC -x c++ -c api_dispatch.cpp -o build/api_dispatch.o; fi
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo "built $OUT"
