#!/bin/bash
# Round 6, GPU call N: is the slow 4th variant of the interleaved A/B tool (r06_ab_call7 / call9) the hardware-queue assignment of the library's second stream?
# Four byte-identical COPIES of the shipped library as four variants, with the runtime's default number of hardware queues and with GPU_MAX_HW_QUEUES=8.
# (First run: the library before the probe of api.hip: aux_stream -> profiles/r06_ab_call10_hw_queues.txt; second run: with the probe.)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06n; rm -rf $O; mkdir -p $O
L=gaussian-opacity-fields_amd/lib
for t in c1 c2 c3; do cp $L/libgof_hip.so $L/libgof_hip_$t.so; done
AB_SCENES=S1M timeout 600 python tests/devtools/dev_r6_ab.py a: b:c1 c:c2 d:c3 > $O/ab_default.txt 2> $O/ab.err; tail -2 $O/ab.err
GPU_MAX_HW_QUEUES=8 AB_SCENES=S1M timeout 600 python tests/devtools/dev_r6_ab.py a: b:c1 c:c2 d:c3 > $O/ab_q8.txt 2> $O/ab.err; tail -2 $O/ab.err
echo "--- default"; cut -c1-420 $O/ab_default.txt; echo "--- GPU_MAX_HW_QUEUES=8"; cut -c1-420 $O/ab_q8.txt
rm -f $L/libgof_hip_c1.so $L/libgof_hip_c2.so $L/libgof_hip_c3.so
