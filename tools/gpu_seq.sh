#!/bin/bash
# Run GPU steps one after another; stop at the first step that was killed or timed out (exit 124 / 137 / >= 128),
# but carry on after an ordinary failure (a failed assertion is information, a hung kernel is not to be followed by more work).
# usage: tools/gpu_seq.sh "cmd1" "cmd2" ...
rc_all=0
for cmd in "$@"; do
    echo "=== $cmd" >&2
    bash -c "$cmd"
    rc=$?
    echo "=== exit $rc" >&2
    if [ $rc -ge 124 ]; then echo "stopping: step killed or timed out" >&2; exit $rc; fi
    [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
