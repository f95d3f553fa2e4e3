#!/bin/bash
# Builds libmla_hip.so for gfx950 (cross-compiles without a GPU). Usage: build.sh [outdir]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/..}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC ${MLA_EXTRA_FLAGS:-} -Wno-unused-value -Wno-unused-result"
mkdir -p "$HERE/build"
pids=()
SRCS="api gemm gemm256 gemm_asm transpose elementwise attention loss pointcloud vision gen"
for f in $SRCS; do
  [ -f "$HERE/$f.hip" ] || continue
  if [ ! -f "$HERE/build/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/build/$f.o" ] || [ "$HERE/common.h" -nt "$HERE/build/$f.o" ] || [ "$HERE/gemm_args.h" -nt "$HERE/build/$f.o" ] || { [ "$f" = gemm_asm ] && { [ "$HERE/gemm_asm_8w_loop.inc" -nt "$HERE/build/$f.o" ] || [ "$HERE/gemm_asm_4w_loop.inc" -nt "$HERE/build/$f.o" ] || [ "$HERE/gemm_asm_8w_clobbers.inc" -nt "$HERE/build/$f.o" ] || [ "$HERE/gemm_asm_4w_clobbers.inc" -nt "$HERE/build/$f.o" ]; }; }; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/build/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
# explicit object list: a stale build/<removed source>.o must not be linked
OBJS=()
for f in $SRCS; do [ -f "$HERE/$f.hip" ] && OBJS+=("$HERE/build/$f.o"); done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libmla_hip.so" "${OBJS[@]}"
echo "built $OUT/libmla_hip.so"
