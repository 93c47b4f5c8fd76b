#!/bin/bash
# Device assembly of the free-gas translation units, for before/after comparisons of a refactoring
# that must not change the generated code:  tools/isa_dump.sh <outdir>
# (then: diff <(grep -v '^\s*[;.]' a/fast.s) <(grep -v '^\s*[;.]' b/fast.s))
set -e
out=${1:?outdir}; mkdir -p "$out"
src=$(dirname "$0")/../ndpp_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNDPP_FAST=1 -ffp-contract=fast -S --cuda-device-only "$src/ndpp_hip.hip" -o "$out/fast.s" &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNDPP_FAST=0 -ffp-contract=off -S --cuda-device-only "$src/fg_strict_stages.hip" -o "$out/strict.s" &
wait
for f in fast strict; do grep -v '^\s*[;.]' "$out/$f.s" | sed 's/;.*$//' > "$out/$f.code"; wc -l "$out/$f.code"; done
