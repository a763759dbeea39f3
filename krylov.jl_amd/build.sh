#!/bin/bash
# Builds libkrylov_hip.so for gfx950 (cross-compiles without a GPU).  -ffp-contract=off: every FMA
# in the kernels is explicit (compensated reductions and bit-exact SpMV depend on it).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/csrc"
OUT="${KHIP_OUT:-$HERE/libkrylov_hip.so}"          # KHIP_OUT / KHIP_BUILD_DIR / KHIP_EXTRA_FLAGS: instrumented variants (tools/archive/spmm_trace.py)
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I/opt/rocm/include $KHIP_EXTRA_FLAGS"
BUILD="${KHIP_BUILD_DIR:-$HERE/build}"
mkdir -p "$BUILD"
objs=""
pids=""
for f in blas1.hip spmv.hip csr_aux.hip spmm_tile.hip panel.hip ilu.hip template.hip colcode.hip coldelta.hip comm.cpp gen_irregular.cpp api.cpp solvers.cpp block.cpp processes.cpp; do
  [ -f "$SRC/$f" ] || continue
  o="$BUILD/${f%.*}.o"
  # spmm_tile.hip: the device assembly is kept beside the object (-save-temps=obj, a by-product of the same compile) --
  # tests/test_build_disasm.py checks the hand-written s_waitcnt vmcnt(N) of its kernels against the VMEM loads hipcc really emitted
  asmf=""; extra=""
  if [ "$f" = "spmm_tile.hip" ]; then asmf="$BUILD/spmm_tile-hip-amdgcn-amd-amdhsa-gfx950.s"; extra="-save-temps=obj"; fi
  if [ ! -f "$o" ] || [ "$SRC/$f" -nt "$o" ] || [ -n "$(find "$SRC" "$HERE/../include" -name '*.h*' -newer "$o" | head -1)" ] || { [ -n "$asmf" ] && [ ! -f "$asmf" ]; }; then
    echo "hipcc $f"
    rm -f "$o"                          # a failed compile must not leave a stale object for the link
    $HIPCC $FLAGS $extra -x hip -c "$SRC/$f" -o "$o" &
    pids="$pids $!"
  fi
  objs="$objs $o"
done
for pid in $pids; do wait $pid || { echo "build.sh: a compile failed" >&2; exit 1; }; done
rm -f "$BUILD"/*.hipi "$BUILD"/*.bc "$BUILD"/*.out "$BUILD"/*.resolution.txt "$BUILD"/*.hipfb "$BUILD"/*-host-*.s "$BUILD"/*-gfx950.o   # -save-temps leftovers (the device .s stays)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" $objs -ldl
echo "built $OUT"
# libkrylov_hip_capi.so: the reference's C / Fortran interface on top (needs ITS header, which is not redistributed)
KH="${KRYLOV_H_DIR:-/root/reference/interfaces/include}"
if [ -n "$KHIP_OUT" ]; then
  exit 0                                  # instrumented variant: the C-interface shim is not rebuilt
fi
if [ -f "$KH/krylov.h" ]; then
  g++ -O2 -fPIC -shared -std=c++17 -I"$KH" -I"$HERE/../include" -o "$HERE/libkrylov_hip_capi.so" "$SRC/capi_compat.cpp" \
      -L"$HERE" -lkrylov_hip -Wl,-rpath,'$ORIGIN'
  echo "built $HERE/libkrylov_hip_capi.so"
else
  echo "krylov.h not found (KRYLOV_H_DIR): libkrylov_hip_capi.so not rebuilt"
fi
