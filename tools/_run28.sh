cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in libtpamd.so libtpamd_nocurve.so; do
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib piped', d['value'], d['ms_per_step'], d['roofline']['kernels']['k_sweep']['ms'])"
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib unpiped', d['value'], d['ms_per_step'], d['roofline']['kernels']['k_sweep']['ms'])"
done; done
for lib in libtpamd.so libtpamd_nocurve.so; do
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 400 python tools/pmc_probe.py r03_curve_pmc_$lib "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" -- --no-pipeline --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_curve_pmc2_$lib.log 2>&1; grep -A14 "^k_sweep" gpurun_out/r03_curve_pmc2_$lib.log
done
find gpurun_out -name "*.db" -delete
DIAG_SO=libtpamd_diag.so timeout -k 10 300 python tools/gpu_diag.py > gpurun_out/r03_curve_diag2.log 2>&1; grep -E "chain-block cycles|scalar-step cycles|init_carry cycles|whole kernel minus|chain steps acc" gpurun_out/r03_curve_diag2.log | cut -c1-150
