#!/bin/bash
# full GPU suite + bench lines (round 2, after the training path)
mkdir -p gpurun_out/full
python -m pytest tests -q -m gpu -x > gpurun_out/full/pytest.log 2>&1
tail -3 gpurun_out/full/pytest.log
python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
cat gpurun_out/full/bench.json
EPRECON_BENCH_FORCE_DIST=1 python bench.py --workload train --steps 8 --warmup 3 2> gpurun_out/full/train.err | grep -v "NCCL\|RCCL" | tail -1 > gpurun_out/full/train.json
cat gpurun_out/full/train.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full/smoke.log 2>&1; tail -2 gpurun_out/full/smoke.log
