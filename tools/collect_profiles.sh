#!/bin/bash
# Round profile collection on the GPU box: kernel stats of the default bench command + PMC passes
# (each counter set in its own run, --kernel-trace only) -> gpurun_out/final2/
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  d=$O/pmc_$(echo $c | cut -c1-18 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $d.log 2>&1
done
ls -R $O | head -40
