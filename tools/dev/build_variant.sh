#!/bin/bash
# usage: build_variant.sh <tag> [extra hipcc flags, e.g. -DJB_TIMING -DJB_DDMC_ALL_WAVES_PER_SIMD=5]
# Builds the working tree's library into variants/libjb_<tag>.so (git-ignored; the A/B scripts load it through
# JAYBENNE_AMD_LIB).  Several of these can run side by side, one core each.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
tag=$1; shift
mkdir -p $ROOT/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics \
  -Wall -Wno-unused-function "$@" -shared $ROOT/jaybenne_amd/csrc/jb_api.hip -o $ROOT/variants/libjb_$tag.so
echo "built variants/libjb_$tag.so $*"
