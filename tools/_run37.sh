cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/libtpamd_noupper.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "repeated" > gpurun_out/r03_dbg_noupper.log 2>&1; echo noupper rc=$?; tail -2 gpurun_out/r03_dbg_noupper.log
grep -q passed gpurun_out/r03_dbg_noupper.log || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "repeated" > gpurun_out/r03_dbg_upper.log 2>&1; echo upper rc=$?; grep -v "^  File\|^\s*$" gpurun_out/r03_dbg_upper.log | head -12
