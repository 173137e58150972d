cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python tools/gpu_fuzz.py 240 301 > gpurun_out/r03_fuzz_301.json 2> gpurun_out/r03_fuzz_301.err; cat gpurun_out/r03_fuzz_301.json
timeout -k 10 400 python tools/gpu_fuzz.py 240 302 > gpurun_out/r03_fuzz_302.json 2> gpurun_out/r03_fuzz_302.err; cat gpurun_out/r03_fuzz_302.json
