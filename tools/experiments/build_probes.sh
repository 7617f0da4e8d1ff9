#!/bin/bash
# Experiment build: every source compiled with -DR3M_PROBES (environment switches + timing probes of DESIGN.md §5b) into
# r3m_amd/lib/variants/libr3m_hip_probes.so — select it with R3M_HIP_LIB=<that path>. The shipped library (csrc/build.sh) has neither.
# usage: [TAG=name] build_probes.sh [extra hipcc flags]      (TAG: separate object dir and library name, e.g. TAG=rev3 ... -DR3M_BN_REV=3)
set -e
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
SRC="$ROOT/r3m_amd/csrc"; SFX="${TAG:+_$TAG}"; OBJ="$ROOT/build/obj_probes$SFX"; OUT="$ROOT/r3m_amd/lib/variants"; mkdir -p "$OBJ" "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DR3M_PROBES $*"
pids=()
for f in conv conv_bf16 stem_bf16 bn loss adam lang augment engine capi; do
  stale=0; [ -f "$OBJ/$f.o" ] || stale=1
  for dep in "$SRC/$f.hip" "$SRC"/*.h "$ROOT/include/r3m_hip.h"; do [ "$dep" -nt "$OBJ/$f.o" ] && stale=1; done
  if [ $stale = 1 ]; then
    /opt/rocm/bin/hipcc $FLAGS -c "$SRC/$f.hip" -o "$OBJ/$f.o" & pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libr3m_hip_probes$SFX.so" "$OBJ"/*.o
echo "built $OUT/libr3m_hip_probes.so"
