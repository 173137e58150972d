cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_i_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_i_pytest.log
tail -3 gpurun_out/r03_i_pytest.log
grep -q " passed" gpurun_out/r03_i_pytest.log || exit 1
timeout -k 10 400 python tools/gpu_other_configs.py 1 2 3 4 > gpurun_out/r03_i_other.jsonl 2>>gpurun_out/r03_i.err; cat gpurun_out/r03_i_other.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'case' in d: print(d['case'], d['ms_per_batch'], d['paths_per_s'], d['bit_exact_on_sample'], d.get('kernels_ms'))"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
