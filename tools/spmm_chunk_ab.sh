#!/bin/bash
# Tile SpMM product-loop variants on ONE box (round 6): the library against instrumented builds of it
# (krylov.jl_amd/build_<tag>/libkrylov_hip.so: c1 = the entry-by-entry loop (the library's since the end of round 6), c2 = chunks of 2 at
# 4 waves per SIMD, c4 = chunks of 4, c4w4 = chunks of 4 forced to 4 waves per SIMD), tools/spmm_ahead_ab.py for each.  KHIP_OUT / KHIP_BUILD_DIR / KHIP_EXTRA_FLAGS of
# krylov.jl_amd/build.sh make the builds.
export TMPDIR=/tmp
for v in "" build_c4 build_c4w4 build_c2 build_c1; do
  [ -z "$v" ] || [ -f krylov.jl_amd/$v/libkrylov_hip.so ] || continue
  if [ -n "$v" ]; then export KHIP_LIBRARY=$PWD/krylov.jl_amd/$v/libkrylov_hip.so; else unset KHIP_LIBRARY; fi
  echo "=== chunk variant: ${v:-library}"
  if [ -z "$v" ]; then timeout 300 python tools/spmm_ahead_ab.py --sweep 2>&1 | cut -c1-175; else timeout 200 python tools/spmm_ahead_ab.py 2>&1 | cut -c1-175; fi
done
unset KHIP_LIBRARY
