#!/bin/bash
# A/B builds for same-box comparisons (gitignored *.so; they travel with gpurun):
#   r3m_amd/lib/libr3m_hip_base.so   = csrc/ of a git ref (default HEAD)             -> R3M_HIP_LIB=... python bench.py
#   r3m_amd/lib/libr3m_hip_probes.so = the working tree with -DR3M_PROBES (environment switches live)
#   r3m_amd/lib/libr3m_hip_<name>.so = the working tree with extra compiler flags (compile-time experiment switches)
# usage: tools/build_ab.sh [base-ref|none] [probes | variant <name> <flags...>]
set -e
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
REF="${1:-HEAD}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
build_tree() {   # $1 = source root holding r3m_amd/csrc + include, $2 = output .so, $3.. = extra flags
  local src="$1" out="$2"; shift 2
  local obj; obj="$(mktemp -d)"
  local pids=()
  for f in conv conv_pw wgrad_win conv_bf16 conv_row16 conv_pw16 stem_bf16 bn loss adam lang augment engine capi; do
    [ -f "$src/r3m_amd/csrc/$f.hip" ] || continue
    local extra=""; [ "$f" = conv_pw16 ] && extra="-mllvm -amdgpu-atomic-optimizer-strategy=None"
    $HIPCC $FLAGS $extra "$@" -c "$src/r3m_amd/csrc/$f.hip" -o "$obj/$f.o" & pids+=($!)
  done
  for p in "${pids[@]}"; do wait "$p"; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out" "$obj"/*.o
  rm -rf "$obj"; echo "built $out"
}
if [ "$REF" != "none" ]; then
  T="$(mktemp -d)"; (cd "$ROOT" && git archive "$REF" r3m_amd/csrc include | tar -x -C "$T")
  build_tree "$T" "$ROOT/r3m_amd/lib/libr3m_hip_base.so"; rm -rf "$T"
fi
if [ "$2" = "probes" ]; then build_tree "$ROOT" "$ROOT/r3m_amd/lib/libr3m_hip_probes.so" -DR3M_PROBES; fi
if [ "$2" = "variant" ]; then N="$3"; shift 3; build_tree "$ROOT" "$ROOT/r3m_amd/lib/libr3m_hip_$N.so" "$@"; fi
