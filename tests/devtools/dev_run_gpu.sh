cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/v13
rm -rf $O; mkdir -p $O
python bench.py 2>/dev/null | tail -1 > $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-loop > $O/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python $R/tests/devtools/dev_pmc.py > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python $R/tests/devtools/dev_pmc.py > $O/pmc_w.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_v -- python $R/tests/devtools/dev_pmc.py > $O/pmc_v.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete
cut -c1-200 $O/bench.json
