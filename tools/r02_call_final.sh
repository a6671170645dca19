#!/bin/bash
# round-2 final: full GPU suite, smoke, bench lines (cfg2 default, cfg4, train), kernel stats of the default bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_final2
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1
grep "passed\|failed" $O/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; grep -i "smoke" $O/smoke.log | tail -1
timeout 600 python bench.py > $O/bench_stdout.txt 2> $O/bench.err; grep '^{' $O/bench_stdout.txt > $O/bench.json
timeout 300 python bench.py --workload cfg4 --steps 24 --warmup 8 2> $O/bench_cfg4.err | grep '^{' > $O/bench_cfg4.json
timeout 300 python bench.py --workload train --steps 8 --warmup 3 2> $O/bench_train.err | grep '^{' > $O/bench_train.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/stats.log 2>&1
rm -f $O/stats/r_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o r -- python $R/bench.py --workload cfg4 --steps 12 --warmup 4 > $O/stats_cfg4.log 2>&1
cd $R
python tools/summarize_cfg4.py $O/stats_cfg4 $O/profiles_r02 $O/bench_cfg4.json > /dev/null 2>&1
rm -f $O/stats_cfg4/r_kernel_trace.csv
python - $O <<'PY'
import json, sys
o = sys.argv[1]
d = json.load(open(o + "/bench.json"))
print("cfg2", round(d["value"], 1), round(d["ms_per_step"], 3), "roofline", round(d["roofline"]["frac"], 3), "conv", round(d["roofline_conv"]["frac"], 3),
      {k: round(v, 2) for k, v in d["extra"].items() if isinstance(v, float)})
print("cfg4", json.load(open(o + "/bench_cfg4.json"))["ms_per_step"], "train", json.load(open(o + "/bench_train.json"))["ms_per_step"])
PY
bash tools/r02_train_prof.sh > gpurun_out/r02_final2/train_prof.txt 2>&1; head -3 gpurun_out/r02_final2/train_prof.txt
