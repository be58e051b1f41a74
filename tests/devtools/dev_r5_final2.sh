#!/bin/bash
# Round 5, second session, last GPU call: the default bench line with the counter passes of THESE kernels committed (the line of the
# evidence pass was taken while the old passes were -- its roofline.traffic is null), the two-ranks-on-one-GPU flow check of the N > 1
# path (gloo), and the opacity-field query's kernel statistics.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/prof5c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-300 $O/bench.json
GOF_BENCH_SHARE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 300 --warmup 5 > $O/bench_n2_shared.json 2> $O/bench_n2.err; tail -2 $O/bench_n2.err; cut -c1-400 $O/bench_n2_shared.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_integrate -- python $GRAFT_REPO_ROOT/tests/devtools/dev_integrate_cache_bench.py > $O/integrate_cache_bench.txt 2> $O/stats_integrate.err ) || tail -3 $O/stats_integrate.err
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
cat $O/integrate_cache_bench.txt | cut -c1-250
