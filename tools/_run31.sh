cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for so in libtpamd_light_curve.so libtpamd_light_nocurve.so; do
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$so timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$so unpiped', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
DIAG_SO=$so timeout -k 10 300 python tools/gpu_diag.py > gpurun_out/r03_light_$so.log 2>&1; cut -c1-150 gpurun_out/r03_light_$so.log | grep -vE " 0 +0 +0 \| +0 +0 +0$"
done
