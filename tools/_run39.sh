cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export AMD_LOG_LEVEL=1
timeout -k 10 400 python tools/gpu_fuzz.py 200 311 > gpurun_out/r03_fuzz_311.json 2> gpurun_out/r03_fuzz_311.err; echo rc=$?; cat gpurun_out/r03_fuzz_311.json | cut -c1-300; grep -v "^\[fuzz\]" gpurun_out/r03_fuzz_311.err | grep -v "^\s*$" | head -20
