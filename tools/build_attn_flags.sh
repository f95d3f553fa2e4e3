#!/bin/bash
# A/B builds of libmla_hip with extra -D flags for attention.hip only. Usage: build_attn_flags.sh <name> "<flags>" [<name> "<flags>" ...]
# Output: mla_amd/csrc/build_tr/lib_<name>.so (git-ignored); run with MLA_HIP_LIB=<that>
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"; C="$HERE/mla_amd/csrc"; mkdir -p "$C/build_tr"
[ -f "$C/build/api.o" ] || bash "$C/build.sh" >/dev/null
while [ $# -ge 2 ]; do
  name="$1"; flags="$2"; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $flags -c "$C/attention.hip" -o "$C/build_tr/attention_$name.o"
  OBJS=(); for o in "$C"/build/*.o; do [ "$(basename "$o")" = attention.o ] || OBJS+=("$o"); done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$C/build_tr/lib_$name.so" "${OBJS[@]}" "$C/build_tr/attention_$name.o"
  rm -f "$C/build_tr/attention_$name.o"; echo "built lib_$name.so"
done
