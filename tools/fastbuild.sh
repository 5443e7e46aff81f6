#!/bin/bash
# Incremental build for iteration: one object per .hip source under /tmp/kzobj (rebuilt when the source or a header is newer),
# linked into kanzi_amd/libkanzi_hip.so.  `python __graft_entry__.py` stays the canonical build (it records the content hash).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=${KZ_OBJ:-/tmp/kzobj}
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $KZ_EXTRA_FLAGS"
cd "$ROOT/kanzi_amd/csrc"
newest_h=$(ls -t *.h ../../include/kanzi_hip.h | head -1)
pids=()
for s in *.hip; do
  o="$OBJ/${s%.hip}.o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ "$newest_h" -nt "$o" ]; then
    ( /opt/rocm/bin/hipcc $FLAGS -c "$s" -o "$o" 2> "$OBJ/${s%.hip}.log" || { cat "$OBJ/${s%.hip}.log"; exit 1; } ) &
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
[ $rc -eq 0 ] || { echo "BUILD FAILED"; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/kanzi_amd/libkanzi_hip.so" "$OBJ"/*.o
echo "built ${#pids[@]} objects"
