#!/bin/bash
# Round 6, R6.12: the evidence for the merged front launch in one session -> gpurun_out/r06_front_evidence.txt
#   needs abtest/front_stamps (bash tools/build_variant.sh front_stamps "-DFRONT_STAMPS" label_head.hip) and abtest/coreside
#   (hipcc -O3 --offload-arch=gfx950 -o abtest/coreside tools/probes/coreside_probe.hip)
cd /root/repo; G=$PWD/gpurun_out; O=$G/r06_front_evidence.txt; : > $O
echo "## 1. co-residency of two 1024-thread workgroups (tools/probes/coreside_probe.hip)" >> $O
./abtest/coreside >> $O 2>&1
echo >> $O; echo "## 2. workgroup timeline of one vrnn_front_kernel launch, configuration 3 (tools/front_timeline.py, -DFRONT_STAMPS build)" >> $O
CLV_LIB=$PWD/abtest/front_stamps/libclvae_hip.so python tools/front_timeline.py 2>&1 | grep -v amdgpu.ids >> $O
echo >> $O; echo "## 3. configuration-3 step, ms, alternating CLV_FRONT_FUSED=1 (one launch) / 0 (label launch + projection launch)" >> $O
for i in 1 2 3; do for V in 1 0; do echo -n "CLV_FRONT_FUSED=$V " >> $O; CLV_FRONT_FUSED=$V python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])" >> $O; done; done
echo >> $O; echo "## 4. rocprofv3 --kernel-trace --stats of the replayed step, both positions" >> $O
for V in 1 0; do
(cd /tmp && CLV_FRONT_FUSED=$V TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $G/r06_front_prof_$V -o p --output-format csv -- python3 /root/repo/bench.py --workload cfg3 --steps 100 --warmup 5 --no-cpu-baseline --no-pmc-traffic --no-also > $G/r06_front_prof.log 2>&1)
python3 - $V <<'PY' >> $O
import csv, sys
for r in list(csv.DictReader(open('/root/repo/gpurun_out/r06_front_prof_%s/p_kernel_stats.csv' % sys.argv[1])))[:10]:
    print('CLV_FRONT_FUSED=%s %-60s %5s calls  %9.1f us' % (sys.argv[1], r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
echo >> $O; echo "## 5. the same two launches as parallel branches of the graph / as two eager streams (tools/sessions/r06_fork*.sh, earlier in the round):" >> $O
echo "   graph branches: 0.3507 / 0.3497 / 0.3505 ms against 0.3428 / 0.3481 / 0.3565 serial; DEBUG_HIP_FORCE_GRAPH_QUEUES=2|4, DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: 0.350-0.361" >> $O
echo "   two eager streams (kernel trace): vrnn_label_fwd_x 0.0-33.1 us, sparse_proj 5.7-36.2 us (alone: 24.7 and 22.1 us)" >> $O
cat $O
