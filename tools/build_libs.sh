#!/bin/bash
# Build the product and diagnostic engine libraries for gfx950 (what engine.build_library() does).
set -e
cd "$(dirname "$0")/../x-edr-trajectory-planning_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17"
hipcc $FLAGS -o libtpamd.so tpamd_capi.hip &
P1=$!
if [ "$1" != "nodiag" ]; then hipcc $FLAGS -DTPAMD_DIAG -o libtpamd_diag.so tpamd_capi.hip & P2=$!; fi
wait $P1
if [ -n "$P2" ]; then wait $P2; fi
echo built
