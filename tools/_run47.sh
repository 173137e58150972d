cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export AMD_LOG_LEVEL=1
for rep in 1 2 3; do
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q > gpurun_out/r03_k_pytest_$rep.log 2>&1; rc=$?; tail -1 gpurun_out/r03_k_pytest_$rep.log
if [ $rc -ne 0 ]; then grep -v "^  File" gpurun_out/r03_k_pytest_$rep.log | grep -v "^\s*$" | head -30; exit 1; fi
done
timeout -k 10 400 python tools/gpu_fuzz.py 240 321 > gpurun_out/r03_fuzz_321.json 2> gpurun_out/r03_fuzz_321.err; echo rc=$?; cut -c1-200 gpurun_out/r03_fuzz_321.json
timeout -k 10 400 python tools/gpu_fuzz.py 240 322 > gpurun_out/r03_fuzz_322.json 2> gpurun_out/r03_fuzz_322.err; echo rc=$?; cut -c1-200 gpurun_out/r03_fuzz_322.json
