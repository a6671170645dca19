#!/bin/bash
# one optimisation step of the 3D path: wall time, kernel list (rocprofv3) and host-side cProfile
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r04_train}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R python $R/tools/profile_train_step.py 10 2>/dev/null | tail -1
PYTHONPATH=$R rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python $R/tools/profile_train_step.py 6 > $O/stats.log 2>&1
rm -f $O/stats/r_kernel_trace.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats/r_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6, "launches", sum(int(r["Calls"]) for r in rows), "(3 + 6 train steps + 6 inference forwards)")
for r in rows[:50]:
    print(f'{r["Name"][:110]:110s} {r["Calls"]:>6s} {float(r["TotalDurationNs"])/1e6:8.3f} ms {float(r["AverageNs"])/1e3:8.1f} us')
PY
PYTHONPATH=$R python - <<PY 2>/dev/null
import cProfile, pstats, torch, os
from eprecon_amd.fragment_step import TrainStep
s = TrainStep(seed=0)
for _ in range(3): s.run()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): s.run()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr)
rows = [(tt/5*1e3, ct/5*1e3, nc/5, f"{os.path.basename(fn)}:{line}:{name}") for (fn,line,name),(cc,nc,tt,ct,_) in st.stats.items()]
rows.sort(reverse=True)
print("# host: own ms/step | cumulative | calls | function")
for r in rows[:45]: print(f"{r[0]:8.3f} {r[1]:8.3f} {r[2]:8.1f}  {r[3]}")
PY
