#!/bin/bash
# Builds libmrec_hip.so for gfx950 (cross-compiles without a GPU).  -ffp-contract=off keeps fp32
# arithmetic bit-identical to the CPU oracle (oracle/Makefile uses the same flag).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRCS=$(ls mrec_*.hip)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fvisibility=hidden \
    -Wall -Wno-unused-function $EXTRA_FLAGS -o ${OUT:-libmrec_hip.so} $SRCS
