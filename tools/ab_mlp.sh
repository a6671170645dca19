set -x
cd /root/repo
timeout 600 python -m pytest tests/test_back_project_gpu.py -x -q 2>&1 | tail -5
for m in 0 1 2 3 4; do for l in 2 1 0; do EPRECON_BP_MLP=$m LVL=$l timeout 120 python tools/ab_backproject.py 2>&1 | tail -1; done; done
