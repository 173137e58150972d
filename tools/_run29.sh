cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in libtpamd.so libtpamd_nocurve.so; do
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib unpiped', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done; done
for lib in libtpamd.so libtpamd_nocurve.so; do
mkdir -p gpurun_out/r03_curve_kt_$lib
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r03_curve_kt_$lib -o kt --output-format csv -- python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-pipeline > /dev/null 2>&1
f=$(find gpurun_out/r03_curve_kt_$lib -name "*kernel_stats.csv" | head -1); echo $lib; cut -d, -f1-6 $f | cut -c1-160 | head -5
done
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*kernel_trace.csv" -delete
