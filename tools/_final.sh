#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05_suite
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05_suite/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05_suite/pytest.log
grep -E "passed|failed" gpurun_out/r05_suite/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()"
bash tools/collect_profiles_r05.sh > gpurun_out/r05_suite/collect.log 2>&1; tail -3 gpurun_out/r05_suite/collect.log
