#!/bin/bash
# Builds libmla_hip.so for gfx950 (cross-compiles without a GPU). Usage: build.sh [outdir]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/..}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# MLA_EXPERIMENTAL=1 adds the opt-in experiment kernels (gemm_asm.hip: assembly main loops, gemm256p: persistent walk, attention_exp.inc:
# the five-product and the one-workgroup-per-head attention backward); the product
# library is built without them. Objects of the two flavours live in separate directories.
EXP="${MLA_EXPERIMENTAL:-0}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC ${MLA_EXTRA_FLAGS:-} -Wno-unused-value -Wno-unused-result"
BUILD="$HERE/build"
if [ "$EXP" = 1 ]; then FLAGS="$FLAGS -DMLA_EXPERIMENTAL_KERNELS"; BUILD="$HERE/build_exp"; fi
mkdir -p "$BUILD"
# identity of the gemm256 kernel family's sources -> mla_gemm_source_id() (api.hip); api.o is rebuilt when it changes
GID=$(cat "$HERE/gemm256.hip" "$HERE/gemm256_kloop.inc" "$HERE/gemm256_kloop_half1.inc" "$HERE/gemm256_kloop_clobbers.inc" "$HERE/gemm_args.h" "$HERE/common.h" | sha256sum | cut -c1-16)
if [ "$(cat "$BUILD/gemm_src_id.txt" 2>/dev/null)" != "$GID" ]; then echo "$GID" > "$BUILD/gemm_src_id.txt"; rm -f "$BUILD/api.o"; fi
pids=()
SRCS="api gemm gemm256 transpose elementwise attention loss pointcloud vision gen contention calib infer"
[ "$EXP" = 1 ] && SRCS="$SRCS gemm_asm"
for f in $SRCS; do
  [ -f "$HERE/$f.hip" ] || continue
  if [ ! -f "$BUILD/$f.o" ] || [ "$HERE/$f.hip" -nt "$BUILD/$f.o" ] || [ "$HERE/common.h" -nt "$BUILD/$f.o" ] || [ "$HERE/gemm_args.h" -nt "$BUILD/$f.o" ] || { [ "$f" = gemm256 ] && { [ "$HERE/gemm256_kloop.inc" -nt "$BUILD/$f.o" ] || [ "$HERE/gemm256_kloop_half1.inc" -nt "$BUILD/$f.o" ] || [ ! -f "$BUILD/$f.s" ]; }; } || { [ "$f" = attention ] && { [ "$HERE/attn_fwd32p_tile0.inc" -nt "$BUILD/$f.o" ] || [ "$HERE/attn_fwd32p_tile1.inc" -nt "$BUILD/$f.o" ] || [ "$HERE/attn_fwd32p_tile0c.inc" -nt "$BUILD/$f.o" ] || [ "$HERE/attn_fwd32p_tile1c.inc" -nt "$BUILD/$f.o" ] || [ "$HERE/attn_fwd32p_clobbers.inc" -nt "$BUILD/$f.o" ] || [ "$HERE/attention_exp.inc" -nt "$BUILD/$f.o" ] || [ ! -f "$BUILD/$f.s" ]; }; } || { [ "$f" = gemm_asm ] && { [ "$HERE/gemm_asm_8w_loop.inc" -nt "$BUILD/$f.o" ] || [ "$HERE/gemm_asm_4w_loop.inc" -nt "$BUILD/$f.o" ] || [ "$HERE/gemm_asm_8w_clobbers.inc" -nt "$BUILD/$f.o" ] || [ "$HERE/gemm_asm_4w_clobbers.inc" -nt "$BUILD/$f.o" ]; }; }; then
    XF=""; [ "$f" = api ] && XF="-DMLA_GEMM_SRC_ID=\"$GID\""
    $HIPCC $FLAGS $XF -c "$HERE/$f.hip" -o "$BUILD/$f.o" &
    pids+=($!)
    if [ "$f" = gemm256 ] || [ "$f" = attention ]; then   # device assembly for tools/check_kloop_asm.py / check_attn_asm.py (below)
      $HIPCC $FLAGS --cuda-device-only -S "$HERE/$f.hip" -o "$BUILD/$f.s" &
      pids+=($!)
    fi
  fi
done
for p in "${pids[@]}"; do wait $p; done
# gemm256's assembly-loop kernels keep their accumulators in a0..a127 across two inline-asm statements: prove from the device assembly
# that the compiler touches no accumulation register (and spills nothing) outside the assembly
python3 "$HERE/../../tools/check_kloop_asm.py" "$BUILD/gemm256.s"
# the opt-in assembly attention forward keeps accumulators, Q and the next tile's scores in physical registers between its statements
python3 "$HERE/../../tools/check_attn_asm.py" "$BUILD/attention.s"
# explicit object list: a stale build/<removed source>.o must not be linked
OBJS=()
for f in $SRCS; do [ -f "$HERE/$f.hip" ] && OBJS+=("$BUILD/$f.o"); done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libmla_hip.so" "${OBJS[@]}"
echo "built $OUT/libmla_hip.so"
