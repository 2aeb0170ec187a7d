#!/bin/bash
# Round 6: the solver step's forms, its stamps, the LM loop's three drivers and the kernel timeline of the library's loop -> gpurun_out/r06/
set -u
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 600 python3 scripts/r06/solver_step_time.py auto,natural 2> $O/solver_step_time.err | grep '^{' > $O/solver_step_time.jsonl; tail -2 $O/solver_step_time.jsonl | cut -c1-300
timeout 300 python3 scripts/r06/solver_forms.py 2> $O/solver_forms.err | grep '^{' > $O/solver_forms.jsonl; cat $O/solver_forms.jsonl | cut -c1-400
for f in True lone-waves teams; do timeout 120 python3 scripts/r06/solver_trace.py $f 2>/dev/null | grep '^{' | grep '"poses": 64'; done > $O/solver_trace.jsonl; cut -c1-200 $O/solver_trace.jsonl
timeout 200 python3 scripts/r06/lm_trial_time.py 2> $O/lm_trial_time.err | grep '^{' > $O/lm_trial_time.jsonl; cat $O/lm_trial_time.jsonl | cut -c1-200
timeout 300 bash scripts/r06/lm_trial_prof.sh > $O/lm_trial_prof.out 2>&1; cp gpurun_out/lm_trial_kernel_stats.csv $O/lm_trial_kernel_stats.csv; cp gpurun_out/lm_trial_timeline.txt $O/lm_trial_timeline.txt; tail -12 $O/lm_trial_timeline.txt
