#!/bin/bash
# builds library variants that differ in conv_bf16.hip compile flags: r3m_amd/lib/variants/libr3m_hip_<tag>.so  (select with R3M_HIP_LIB)
# usage: build_variants.sh tag1="-DX=1" tag2="-DY=2" ...
set -e
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
OBJ="$ROOT/build/obj"; OUT="$ROOT/r3m_amd/lib/variants"; mkdir -p "$OUT"
bash "$ROOT/r3m_amd/csrc/build.sh" > /dev/null
for spec in "$@"; do
  tag="${spec%%=*}"; flags="${spec#*=}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c "$ROOT/r3m_amd/csrc/conv_bf16.hip" -o "$OBJ/conv_bf16_$tag.o" &
done
wait
for spec in "$@"; do
  tag="${spec%%=*}"
  objs=(); for f in conv stem_bf16 bn loss adam lang augment engine capi; do objs+=("$OBJ/$f.o"); done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libr3m_hip_$tag.so" "${objs[@]}" "$OBJ/conv_bf16_$tag.o"
  echo "built $OUT/libr3m_hip_$tag.so"
done
