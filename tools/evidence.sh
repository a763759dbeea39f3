#!/bin/bash
# tools/evidence.sh <round-tag> <stage ...> -- the GPU passes behind profiles/<round-tag>_* (one script instead of the per-letter
# gpu_r0x_*.sh of earlier rounds).  Run on the GPU box: gpurun --timeout 900 -- 'bash tools/evidence.sh r05 tests bench'
# Stages (any order, each under its own timeout; outputs under gpurun_out/, copy what is to be judged into profiles/):
#   tests     pytest -m gpu (parity log -> gpurun_out/parity_log.jsonl) + smoke()
#   tests_create  the GPU suite with KHIP_PY_WORKSPACES=create (library-owned workspaces)
#   bench     bench.py at the driver's setting (20 steps, 5 warm-up) and at 100 steps
#   floors    tools/streamfloor: 2R / 3R (+1W) panel streams, tile-SpMM twin
#   slab      tools/slab_iteration.py: one rank's CG iteration at the N = 1 / 2 / 4 / 8 slab shapes, self halo over RCCL
#   slabprof  rocprofv3 --kernel-trace --stats of the N = 8 slab iteration
#   prof      rocprofv3 --kernel-trace --stats of bench.py; PMC passes of the SpMV (tools/gpu_prof.sh)
#   cfg5      cfg-5 block-GMRES: kernel stats + bench_configs.py
#   prio      halo-stream priority A/B at the N = 8 slab + kernel trace
#   newtests  the GPU tests added last
#   ilu       tools/bench_ilu.py 64 128 256
#   adopt     tests/c/adopt_sequence 64 512
#   spmmirr   rocprofv3 --pmc passes + synthetic twin of the tile SpMM on the banded + random operator
#   spmmab    tile SpMM variants (tools/spmm_ahead_ab.py --sweep) + the twins of its traffic on the same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; TAG=${1:-r05}; shift; mkdir -p gpurun_out; export TMPDIR=/tmp
for stage in "$@"; do
  echo "=== $stage"
  case $stage in
    tests)
      rm -f gpurun_out/parity_log.jsonl
      timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -25
      timeout 100 python -c "import __graft_entry__ as g; g.smoke()" ;;
    tests_create)      # the same suite on library-owned workspaces (khip_*_workspace_create), the other kind the ABI offers
      KHIP_PY_WORKSPACES=create timeout 1200 python -m pytest tests -q -m gpu --deselect tests/test_gpu_adopt.py 2>&1 | tail -6 ;;
    bench)
      timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_setting.json 2> gpurun_out/${TAG}_bench.err; tail -c 600 gpurun_out/${TAG}_bench_driver_setting.json
      timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2>> gpurun_out/${TAG}_bench.err; tail -c 300 gpurun_out/${TAG}_bench.json ;;
    floors)
      [ -x tools/streamfloor ] && [ tools/streamfloor -nt tools/streamfloor.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/streamfloor tools/streamfloor.hip
      timeout 200 tools/streamfloor panel > gpurun_out/${TAG}_panel_stream_floor.log 2>&1; cat gpurun_out/${TAG}_panel_stream_floor.log
      timeout 300 tools/streamfloor spmm > gpurun_out/${TAG}_spmm_floor.log 2>&1; cat gpurun_out/${TAG}_spmm_floor.log
      timeout 200 tools/streamfloor cgfold > gpurun_out/${TAG}_cgfold_twin.log 2>&1; cat gpurun_out/${TAG}_cgfold_twin.log ;;
    slab)
      rm -f gpurun_out/${TAG}_slab_iteration.jsonl
      timeout 200 python tools/slab_iteration.py --only 1 --out gpurun_out/${TAG}_slab_iteration.jsonl 2>&1 | tail -2
      for N in 2 4 8; do timeout 200 python tools/slab_iteration.py --only $N --variants --out gpurun_out/${TAG}_slab_iteration.jsonl 2>&1 | grep '^{' | cut -c1-200; done ;;
    slabprof)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_slab8_stats -o s -- python $R/tools/slab_iteration.py --only 8 --out $R/gpurun_out/${TAG}_slab8_prof.jsonl > $R/gpurun_out/${TAG}_slab8_prof.log 2>&1; echo "slabprof exit $?")
      head -12 gpurun_out/${TAG}_slab8_stats/s_kernel_stats.csv | cut -c1-220 ;;
    prof)
      bash tools/gpu_prof.sh ${TAG} 2>&1 | tail -20 ;;
    cfg5)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_cfg5_stats -o s -- python $R/tools/cfg5_only.py > $R/gpurun_out/${TAG}_cfg5_prof.log 2>&1; echo "cfg5 prof exit $?")
      head -14 gpurun_out/${TAG}_cfg5_stats/s_kernel_stats.csv | cut -c1-220
      timeout 600 python tools/bench_configs.py > gpurun_out/${TAG}_bench_configs.jsonl 2> gpurun_out/${TAG}_bench_configs.err; cut -c1-400 gpurun_out/${TAG}_bench_configs.jsonl ;;
    prio)
      # halo stream priority A/B on the N = 8 slab (same box): KHIP_COMM_PRIORITY = 1 (default: highest) vs 0
      for P in 1 0 1 0; do KHIP_COMM_PRIORITY=$P timeout 200 python tools/slab_iteration.py --only 8 --out gpurun_out/${TAG}_slab_prio$P.jsonl 2>&1 | grep '^{' | cut -c1-170; done
      (cd /tmp && KHIP_COMM_PRIORITY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_slab8p_stats -o s -- python $R/tools/slab_iteration.py --only 8 --out $R/gpurun_out/${TAG}_slab8p_prof.jsonl > $R/gpurun_out/${TAG}_slab8p_prof.log 2>&1; echo "slab8 priority trace exit $?") ;;
    newtests)
      timeout 900 python -m pytest tests/test_gpu_scale_parity.py::test_block_gmres_cfg5_banded_random_matches_oracle tests/test_gpu_adopt.py tests/test_gpu_self_halo.py -q 2>&1 | tail -8
      grep banded gpurun_out/parity_log.jsonl | tail -1 ;;
    ilu)
      timeout 400 python tools/bench_ilu.py 64 128 256 > gpurun_out/${TAG}_bench_ilu.jsonl 2> gpurun_out/${TAG}_bench_ilu.err; cut -c1-500 gpurun_out/${TAG}_bench_ilu.jsonl ;;
    adopt)
      timeout 300 tests/c/adopt_sequence 64 512 2>&1 | tail -12 ;;
    spmmirr)     # counters + twin of the tile SpMM on the banded + random operator
      bash tools/gpu_prof_spmm_irregular.sh ${TAG} 2>&1 | tail -60 ;;
    spmmab)      # tile SpMM variants on one box (look-ahead, chunked product loop, grids) + the twins of its traffic
      timeout 300 python tools/spmm_ahead_ab.py --sweep > gpurun_out/${TAG}_spmm_ahead_ab.jsonl 2> gpurun_out/${TAG}_spmm_ahead_ab.err; cut -c1-170 gpurun_out/${TAG}_spmm_ahead_ab.jsonl
      [ -x tools/streamfloor ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/streamfloor tools/streamfloor.hip
      (timeout 100 tools/streamfloor spmmslide | grep "waves/CU= 4"; timeout 100 tools/streamfloor spmm | grep "twin:\|entries + records + panel rows + Y waves/CU= 4") 2>&1 | cut -c1-150 | tee gpurun_out/${TAG}_spmm_twins.log ;;
    *) echo "unknown stage $stage" ;;
  esac
done
