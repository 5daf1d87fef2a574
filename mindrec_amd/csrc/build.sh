#!/bin/bash
# Builds libmrec_hip.so for gfx950 (cross-compiles without a GPU).  -ffp-contract=off keeps fp32
# arithmetic bit-identical to the CPU oracle (oracle/Makefile uses the same flag).
# Each mrec_*.hip is compiled to its own object (in parallel, only when stale), then linked.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${OUT:-libmrec_hip.so}
OBJ=${OBJDIR:-.obj}
[ -n "$EXTRA_FLAGS" ] && OBJ="$OBJ-$(echo "$EXTRA_FLAGS" | cksum | cut -d' ' -f1)"
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function $EXTRA_FLAGS"
pids=()
for src in mrec_*.hip; do
    obj="$OBJ/${src%.hip}.o"
    stale=0
    [ -f "$obj" ] || stale=1
    for dep in "$src" mrec_*.h ../../include/mrec.h build.sh; do
        [ "$stale" = 0 ] && [ "$dep" -nt "$obj" ] && stale=1
    done
    if [ "$stale" = 1 ]; then
        $HIPCC $FLAGS -c "$src" -o "$obj" &
        pids+=($!)
    fi
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/mrec_*.o
