#!/bin/bash
# Round-4 iteration check on the GPU box: the GPU test suite (all failures listed, not only the first), the cfg4 stage
# breakdown and a short cfg4 bench (unpipelined) -> gpurun_out/$1/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04_a}
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --maxfail 12 --timeout 600 ${PYTEST_ARGS} > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 600 python tools/profile_cfg4_stages.py 3 > $O/cfg4_stage_times.txt 2>&1
tail -25 $O/cfg4_stage_times.txt
EPRECON_CFG4_PIPELINE=0 timeout 600 python bench.py --workload cfg4 --steps 16 --warmup 8 > $O/bench_cfg4_unpipelined.json 2> $O/bench_cfg4_unpipelined.err
tail -2 $O/bench_cfg4_unpipelined.json | cut -c1-400
