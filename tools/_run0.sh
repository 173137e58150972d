cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/gpu_other_configs.py 4 > gpurun_out/r03_base_cfg4.jsonl 2> gpurun_out/r03_base_cfg4.err &&
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_base_cfg4_prof -o cfg4 --output-format csv -- python3 tools/gpu_other_configs.py 4 > gpurun_out/r03_base_cfg4_prof.log 2>&1 &&
ls gpurun_out/r03_base_cfg4_prof
