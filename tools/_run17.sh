cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
DIAG_HIST=gpurun_out/r03_tail_histogram.json timeout -k 10 300 python tools/gpu_diag.py > gpurun_out/r03_p_diag.log 2>&1; cat gpurun_out/r03_tail_histogram.json | head -30
timeout -k 10 400 python tools/gpu_timeline.py r03_tail_tl x-edr-trajectory-planning_amd/csrc/libtpamd.so > gpurun_out/r03_tail_timeline.log 2>&1; tail -3 gpurun_out/r03_tail_timeline.log
timeout -k 10 400 python tools/gpu_timeline.py r03_tail_tl2 x-edr-trajectory-planning_amd/csrc/libtpamd.so -- --pipeline 2 > gpurun_out/r03_tail_timeline_mode2.log 2>&1; tail -2 gpurun_out/r03_tail_timeline_mode2.log
for b in 1024 2048 4096 8192; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline --paths-per-gpu $b 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('paths', $b, 'unpiped ms/step', d['ms_per_step'], 'per 1024', round(d['ms_per_step']*1024/$b,4), d['roofline']['kernels'])"; done
