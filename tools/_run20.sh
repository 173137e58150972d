cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
C=x-edr-trajectory-planning_amd/csrc
echo "== s_setprio in K1"; timeout -k 10 500 python tools/gpu_timeline.py r03_r_tl $C/libtpamd.so $C/libtpamd_k1prio1.so $C/libtpamd_k1prio3.so 2>&1 | tail -3
echo "== aux stream priority high"; TPAMD_AUX_PRIORITY=high timeout -k 10 300 python tools/gpu_timeline.py r03_r_tl2 $C/libtpamd.so 2>&1 | tail -1
echo "== aux stream priority default"; TPAMD_AUX_PRIORITY=default timeout -k 10 300 python tools/gpu_timeline.py r03_r_tl3 $C/libtpamd.so 2>&1 | tail -1
for t in 64 256; do echo "== K1 tpb $t"; TPAMD_K1_TPB=$t timeout -k 10 300 python tools/gpu_timeline.py r03_r_tl4_$t $C/libtpamd.so 2>&1 | tail -1; done
rm -rf gpurun_out/r03_r_tl*
