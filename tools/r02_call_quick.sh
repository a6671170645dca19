#!/bin/bash
# conv-affected GPU tests + bench lines
mkdir -p gpurun_out/quick
timeout 900 python -m pytest tests/test_sparse_gpu.py tests/test_dense2d_gpu.py tests/test_spvcnn_gpu.py tests/test_occupancy_init_gpu.py tests/test_neucon_gpu.py tests/test_pins_gpu.py tests/test_cfg4_gpu.py tests/test_autograd_gpu.py -q -m gpu -x > gpurun_out/quick/pytest.log 2>&1
grep "passed\|failed" gpurun_out/quick/pytest.log | tail -2
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/quick/bench.json 2> gpurun_out/quick/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/quick/bench.json"))
print("cfg2", d["value"], d["ms_per_step"], "conv", d["roofline_conv"]["avg_launch_ms"], d["roofline_conv"]["frac"], "extra", {k: v for k, v in d["extra"].items() if k.endswith(("ms_per_fragment", "ms_per_step"))})
PY
