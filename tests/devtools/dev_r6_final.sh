#!/bin/bash
# Round 6: the evidence pass on HEAD's kernels (1x MI355X).  Order = what must not be lost first: the counter passes the committed bench
# line quotes (separate --pmc runs, never combined with sys / hip traces), the default bench line, the kernel statistics of the bench
# command and of the 6M-Gaussian leg, the GPU suite WITHOUT -x + smoke, then the remaining counter groups.  Outputs under gpurun_out/prof6/.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/prof6
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; ( cd /tmp && timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc/$tag -- python $GRAFT_REPO_ROOT/tests/devtools/dev_pmc.py > /tmp/pmc_$tag.log 2>&1 ) || tail -5 /tmp/pmc_$tag.log; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-integrate --no-full-loop --no-clustered --no-views --no-reference --no-kernel-size-leg --no-large-p > $O/stats_bench.json 2> $O/stats.err ) || tail -3 $O/stats.err
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats6m -- python $GRAFT_REPO_ROOT/tests/devtools/dev_r6_large_p_trace.py > $O/stats6m.txt 2> $O/stats6m.err ) || tail -3 $O/stats6m.err
timeout 1500 python -m pytest tests -q -m gpu -rf --tb=short --durations=15 > $O/tests_full.txt 2>&1; tail -40 $O/tests_full.txt | cut -c1-1200 > $O/tests.txt; cat $O/tests.txt
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.txt 2>&1; cat $O/smoke.txt
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_WAVES
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run sq4 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
du -sh $O; cat $O/bench.json | cut -c1-600
