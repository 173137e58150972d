cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/gpu_plan_bench.py 1024 7 16 1 3 > gpurun_out/r03_plan_bench.json 2> gpurun_out/r03_plan_bench.err; echo rc=$?; cat gpurun_out/r03_plan_bench.json; tail -3 gpurun_out/r03_plan_bench.err
timeout -k 10 600 python tools/gpu_plan_bench.py 1024 7 16 0 12 > gpurun_out/r03_plan_bench_w12.json 2>> gpurun_out/r03_plan_bench.err; cat gpurun_out/r03_plan_bench_w12.json
