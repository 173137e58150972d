cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_host_api as t
print(t._build_test_binary())
" > gpurun_out/r03_k_build.log 2>&1
timeout -k 10 300 tests/cpp/test_host_api > gpurun_out/r03_k_hostapi.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r03_k_hostapi.log
