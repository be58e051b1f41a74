#!/bin/bash
# Round-5 evidence pass on the GPU box: rocprofv3 kernel statistics of the bench command, the PMC passes (one counter group per run;
# never combined with sys/hip traces), the default bench line, the integrate legs' kernel statistics.  Outputs under gpurun_out/prof5/.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/prof5
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-integrate --no-full-loop --no-clustered --no-views --no-reference --no-kernel-size-leg --no-large-p > $O/stats_bench.json 2> $O/stats.err ) || tail -3 $O/stats.err
run() { tag=$1; shift; ( cd /tmp && timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc/$tag -- python $GRAFT_REPO_ROOT/tests/devtools/dev_pmc.py > /tmp/pmc_$tag.log 2>&1 ) || tail -5 /tmp/pmc_$tag.log; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_WAVES
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run sq4 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
# integrate (config 5 shape): kernel statistics of first + cached calls
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_integrate -- python $GRAFT_REPO_ROOT/tests/devtools/dev_integrate_cache_bench.py > $O/integrate_cache_bench.txt 2> $O/stats_integrate.err ) || tail -3 $O/stats_integrate.err
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
du -sh $O
