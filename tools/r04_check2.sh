#!/bin/bash
# quick iteration: selected tests + decoder timing + cfg4 stage times + short cfg4 bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04_x}
mkdir -p $O
cd $R
python -m pytest ${PYTEST_SEL:-tests/test_mask3dformer.py tests/test_sparse_gpu.py tests/test_spvcnn_gpu.py tests/test_cfg4_gpu.py tests/test_free_run_gpu.py tests/test_neucon_gpu.py} -m gpu -q --maxfail 12 --timeout 600 > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python tools/profile_decoder.py 30 > $O/decoder_times.txt 2>&1; tail -2 $O/decoder_times.txt
timeout 600 python tools/profile_cfg4_stages.py 3 > $O/cfg4_stage_times.txt 2>&1
tail -24 $O/cfg4_stage_times.txt
EPRECON_CFG4_PIPELINE=0 timeout 600 python bench.py --workload cfg4 --steps 16 --warmup 8 > $O/bench_cfg4_unpipelined.json 2> $O/bench_cfg4_unpipelined.err
tail -1 $O/bench_cfg4_unpipelined.json | cut -c1-200

