cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_final3_bench_$rep.json 2> gpurun_out/r03_final3_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03_final3_bench_$rep.json')); print('bench', d['value'], d['ms_per_step'], 'cold', d['cold_value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline'].get('step_hbm_frac'), d['cpu_baseline']['value'], d['config'].get('dominant_kernel_timed_every'))"
done
timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline > gpurun_out/r03_final3_bench_300.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03_final3_bench_300.json')); print('bench300', d['value'], d['ms_per_step'])"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline > gpurun_out/r03_final3_bench_unpiped.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03_final3_bench_unpiped.json')); print('unpiped20', d['value'], d['ms_per_step'])"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipeline 2 > gpurun_out/r03_final3_bench_mode2.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03_final3_bench_mode2.json')); print('mode2', d['value'], d['ms_per_step'])"
