cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/libtpamd_noupper.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_dbg2_noupper.log 2>&1; echo noupper rc=$?; tail -2 gpurun_out/r03_dbg2_noupper.log
grep -q passed gpurun_out/r03_dbg2_noupper.log || exit 1
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -v > gpurun_out/r03_dbg2_upper.log 2>&1; echo upper rc=$?; grep -v "^  File\|^\s*$" gpurun_out/r03_dbg2_upper.log | tail -12
