#!/bin/bash
# Builds libr3m_hip.so for gfx950 (MI355X). hipcc cross-compiles without a GPU. Usage: build.sh [extra hipcc flags]
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/../../build/obj"
OBJ="$HERE/../../build/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*"
pids=()
SRCS="conv conv_pw wgrad_win conv_bf16 conv_row16 conv_pw16 stem_bf16 bn loss adam lang augment engine capi"
for f in $SRCS; do
  [ -f "$HERE/$f.hip" ] || continue
  stale=0
  [ -f "$OBJ/$f.o" ] || stale=1
  for dep in "$HERE/$f.hip" "$HERE"/*.h "$HERE/../../include/r3m_hip.h" "$HERE/build.sh"; do
    [ "$dep" -nt "$OBJ/$f.o" ] && stale=1
  done
  if [ $stale = 1 ]; then
    EXTRA=""
    # conv_pw16: the tile tickets are requested one tile before they are used; the wave-level atomic optimizer would wait for each at once
    [ "$f" = conv_pw16 ] && EXTRA="-mllvm -amdgpu-atomic-optimizer-strategy=None"
    $HIPCC $FLAGS $EXTRA -c "$HERE/$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
objs=()
for f in $SRCS; do [ -f "$OBJ/$f.o" ] && objs+=("$OBJ/$f.o"); done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libr3m_hip.so" "${objs[@]}"
echo "built $OUT/libr3m_hip.so"
