# A/B of two engine libraries built from one tree (TPAMD_LIBRARY selects the .so): parity tests first, then
# 3 x 300 pipelined steps and 300 unpipelined steps of each. Build the variant with
#   engine.build_library(force=True, extra_flags=["-D..."], output=".../libtpamd_prev.so")
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r03_setup_tests.log 2>&1; tail -2 gpurun_out/r03_setup_tests.log
grep -q " failed\|error" gpurun_out/r03_setup_tests.log && exit 1; grep -q " passed" gpurun_out/r03_setup_tests.log || exit 1
for rep in 1 2 3; do for lib in libtpamd.so libtpamd_prev.so; do
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib piped', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done; done
for lib in libtpamd.so libtpamd_prev.so; do
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib unpiped', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done
