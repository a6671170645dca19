#!/bin/bash
# round 5, second GPU call: the suite on the prefetch / fused-reset build, cfg4 figures (ms, reads), kernel statistics of a cfg4 run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05b
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
EPRECON_CFG4_PIPELINE=0 timeout 300 python bench.py --workload cfg4 --steps 24 --warmup 8 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"; python -c "import json;d=json.load(open(\"$O/bench_cfg4.json\"));print(d[\"ms_per_step\"], d[\"blocking_reads_per_fragment\"])"
EPRECON_PREFETCH=0 EPRECON_CFG4_PIPELINE=0 timeout 300 python bench.py --workload cfg4 --steps 24 --warmup 8 > $O/bench_cfg4_noprefetch.json 2> $O/bench_cfg4_noprefetch.err; python -c "import json;d=json.load(open(\"$O/bench_cfg4_noprefetch.json\"));print(d[\"ms_per_step\"], d[\"blocking_reads_per_fragment\"])"
cd /tmp && export TMPDIR=/tmp
EPRECON_CFG4_PIPELINE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o r -- python $R/bench.py --workload cfg4 --steps 12 --warmup 4 > $O/stats_cfg4.log 2>&1
cd $R
mkdir -p $O/profiles
python tools/summarize_cfg4.py $O/stats_cfg4 $O/profiles $O/bench_cfg4.json > /dev/null 2>$O/summarize.err; cat $O/profiles/cfg4_kernel_stats.json
rm -f $O/stats_cfg4/r_kernel_trace.csv $O/stats_cfg4/*/r_kernel_trace.csv
python - <<PY
import json
from eprecon_amd import _lib
PY
