#!/bin/bash
# TIMING-ONLY ablation builds of the assembly attention forward (wrong results): libmla_hip with the generated tile bodies stripped of one
# instruction class (GEN_ATTN_ABL: nosm = no softmax VALU, nomfma, nok / nov = no K / V^T fragment reads, nold = no global -> LDS staging copies). Usage: build_attn_abl.sh nosm nomfma ...
# Output: mla_amd/csrc/build_tr/libabl_<name>.so (git-ignored); run with MLA_HIP_LIB=<that> MLA_ATTN_FWD=1 python tools/bench_attn.py
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"; C="$HERE/mla_amd/csrc"; mkdir -p "$C/build_tr"
[ -f "$C/build/api.o" ] || bash "$C/build.sh" >/dev/null
for abl in "$@"; do
  T=$(mktemp -d /tmp/abl.XXXX)
  for f in "$C"/*.hip "$C"/*.h "$C"/*.inc; do ln -s "$f" "$T/$(basename "$f")"; done
  rm -f "$T"/attn_fwd32*.inc
  XF=""; case "$abl" in *nold*) XF="-DMLA_ATTN_NOSTAGE";; esac
  GEN_OUT_DIR="$T" GEN_ATTN_ABL="$abl" python3 "$HERE/tools/gen_attn_asm.py" >/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $XF $ABL_FLAGS -c "$T/attention.hip" -o "$T/attention.o"
  OBJS=(); for o in "$C"/build/*.o; do [ "$(basename "$o")" = attention.o ] || OBJS+=("$o"); done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$C/build_tr/libabl_${abl//,/_}.so" "${OBJS[@]}" "$T/attention.o"
  rm -rf "$T"; echo "built libabl_${abl//,/_}.so"
done
