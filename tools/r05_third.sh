#!/bin/bash
# round 5, third GPU call: the native SPVCNN pass and the pinned reads — parity tests, then cfg4 A/B in one process each
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_spvcnn_gpu.py tests/test_switches_gpu.py tests/test_cfg4_gpu.py tests/test_free_run_gpu.py tests/test_neucon_gpu.py tests/test_gru_fusion_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -6 $O/pytest.log
run() { # name, env...
  name=$1; shift
  env "$@" EPRECON_CFG4_PIPELINE=0 timeout 300 python bench.py --workload cfg4 --steps 32 --warmup 8 > $O/cfg4_$name.json 2> $O/cfg4_$name.err
  python -c "import json;d=json.load(open('$O/cfg4_$name.json'));print('$name', round(d['ms_per_step'],3), d['blocking_reads_per_fragment'])"
}
run default A=1
run python_spvcnn EPRECON_SPVCNN_NATIVE=0
run tolist_reads EPRECON_PINNED_READS=0
run default2 A=1
run r04_like EPRECON_SPVCNN_NATIVE=0 EPRECON_PINNED_READS=0 EPRECON_PREFETCH=0
