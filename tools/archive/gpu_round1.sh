#!/bin/bash
# First GPU pass: parity tests, smoke, short bench, kernel sweep, rocprofv3 kernel stats.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing|gfx9" | head -4 > gpurun_out/device.txt 2>&1
nproc >> gpurun_out/device.txt; free -g | head -2 >> gpurun_out/device.txt
echo "== pytest gpu" ; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench exit $?"; cat gpurun_out/bench1.json; tail -5 gpurun_out/bench1.err
echo "== sweep"; timeout 600 python tools/archive/sweep_kernels_512.py --quick > gpurun_out/sweep.log 2>&1; echo "sweep exit $?"; tail -40 gpurun_out/sweep.log
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_r1" -o r1 -- python "$OLDPWD/bench.py" --steps 20 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1; echo "rocprof exit $?"; cd "$OLDPWD"
ls -R gpurun_out/prof_r1 | head -20
