cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cf in 1 0; do
TPAMD_CHAIN_FRONTS=$cf TPAMD_BUCKET=0 timeout -k 10 300 python tools/gpu_other_configs.py 4 2>>gpurun_out/r03_c.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain=$cf bucket=0', d['ms_per_batch'], d['paths_per_s'], d['dof_groups_in_turn_ms'], d['buckets_equal_dof_groups_on_every_path'], d['bit_exact_on_sample'])"
done
TPAMD_BUCKET=2560 timeout -k 10 300 python tools/gpu_other_configs.py 4 2>>gpurun_out/r03_c.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bucket=2560 lanes=4', d['ms_per_batch'], d['paths_per_s'])"
GPU_MAX_HW_QUEUES=8 TPAMD_LANES=6 TPAMD_BUCKET=2560 timeout -k 10 300 python tools/gpu_other_configs.py 4 2>>gpurun_out/r03_c.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bucket=2560 lanes=6 hwq=8', d['ms_per_batch'], d['paths_per_s'])"
GPU_MAX_HW_QUEUES=8 TPAMD_LANES=8 TPAMD_BUCKET=1024 timeout -k 10 300 python tools/gpu_other_configs.py 4 2>>gpurun_out/r03_c.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bucket=1024 lanes=8 hwq=8', d['ms_per_batch'], d['paths_per_s'])"
TPAMD_BUCKET=0 timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/r03_c_tl -o b0 --output-format csv -- python3 tools/gpu_other_configs.py 4 > gpurun_out/r03_c_tl.log 2>&1
python tools/kernel_timeline.py gpurun_out/r03_c_tl/b0_kernel_trace.csv 14
DIAG_B=512 DIAG_D=14 DIAG_N=4000 timeout -k 10 300 python tools/gpu_diag.py > gpurun_out/r03_c_diag14.log 2>&1; tail -32 gpurun_out/r03_c_diag14.log
