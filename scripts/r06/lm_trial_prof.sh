#!/bin/bash
# Round 6: kernel trace of the library's LM loop on BASELINE configs[2]'s graph (scripts/r06/lm_trial_time.py) -> profiles/r06_lm_trial_kernel_stats.csv + a per-iteration timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/lm_prof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/lm_prof -o lm --output-format csv -- python $R/scripts/r06/lm_trial_time.py > $R/gpurun_out/lm_prof.log 2>&1
tail -3 $R/gpurun_out/lm_prof.log
find $R/gpurun_out/lm_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/lm_trial_kernel_stats.csv
python3 - <<'P'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/lm_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last 40 kernels: name, duration, gap to the previous one's end
prev = None
out = []
for r in rows[-60:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void gp::", "").replace("(anonymous namespace)::", "")[:60]
    out.append(f"{name:60s} dur {(e - s) / 1e3:8.2f} us  gap {((s - prev) / 1e3 if prev else 0):8.2f} us")
    prev = e
open(R + "/gpurun_out/lm_trial_timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[-24:]))
P
