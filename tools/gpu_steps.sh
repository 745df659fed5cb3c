#!/bin/bash
# tools/gpu_steps.sh -- run the lines of a step file one after the other on the GPU box, each with its own log under the output
# directory; an ordinary failure (a test that fails: status < 124) lets the next step run, a step that was killed or timed out
# (status >= 124) ends the call -- no further GPU step starts after one that hung.
#   usage: bash tools/gpu_steps.sh <outdir> <<'STEPS'
#          name|timeout_seconds|command ...
#          STEPS
out=$1; mkdir -p "$out"
while IFS='|' read -r name tmo cmd; do
    [ -z "$name" ] && continue
    echo "== $name: $cmd" | tee -a "$out/steps.log"
    t0=$(date +%s)
    timeout -k 10 "$tmo" bash -c "$cmd" > "$out/$name.log" 2> "$out/$name.err" < /dev/null      # (mpirun and friends read stdin: the step list is ours)
    rc=$?
    echo "== $name: status $rc after $(( $(date +%s) - t0 )) s" | tee -a "$out/steps.log"
    tail -n 3 "$out/$name.log"
    if [ $rc -ge 124 ]; then echo "== stopping: $name was killed or timed out" | tee -a "$out/steps.log"; exit $rc; fi
done
exit 0
