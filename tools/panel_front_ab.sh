#!/bin/bash
# Panel-kernel build variants against the library on ONE box (tools/panel_nt_ab.py per build: cfg-5 block_gmres!, HIP-event averages per kernel).
# Builds: krylov.jl_amd/build_<tag>/libkrylov_hip.so made with KHIP_OUT / KHIP_BUILD_DIR / KHIP_EXTRA_FLAGS of build.sh:
#   tn64 / tn128  -DKHIP_ROWS_PER_WAVE_TN=64 / 128   rows per wave of the V^T Q fold (library: 256): a tighter access front
#   w4            -DKHIP_NN_TN_WAVES=4                register allocation of panel_nn_tn_kernel capped for 4 waves per SIMD (library: 148 VGPRs = 3)
export TMPDIR=/tmp
for v in "" build_w4 build_tn128 build_tn64 "" build_w4; do
  [ -z "$v" ] || [ -f krylov.jl_amd/$v/libkrylov_hip.so ] || continue
  if [ -n "$v" ]; then export KHIP_LIBRARY=$PWD/krylov.jl_amd/$v/libkrylov_hip.so; else unset KHIP_LIBRARY; fi
  echo "=== ${v:-library}"
  timeout 200 python tools/panel_nt_ab.py 0 2>&1 | tail -1 | cut -c1-400
done
unset KHIP_LIBRARY
