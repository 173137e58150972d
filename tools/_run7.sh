cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_g_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_g_pytest.log
tail -4 gpurun_out/r03_g_pytest.log
run() { tag="$1"; shift; env "$@" timeout -k 10 300 python tools/gpu_other_configs.py 4 2>>gpurun_out/r03_g.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['buckets'], d['ms_per_batch'], d['paths_per_s'], d['dof_groups_in_turn_ms'], d['buckets_equal_dof_groups_on_every_path'], d['bit_exact_on_sample'])"; }
run "cfg4" TPAMD_BUCKET=0
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_g_bench.json 2> gpurun_out/r03_g_bench.err; cat gpurun_out/r03_g_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline'])"
timeout -k 10 300 python tools/gpu_other_configs.py 1 2 3 > gpurun_out/r03_g_other.jsonl 2>>gpurun_out/r03_g.err; cat gpurun_out/r03_g_other.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'case' in d: print(d['case'], d['ms_per_batch'], d['paths_per_s'], d['bit_exact_on_sample'], d['kernels_ms'])"
