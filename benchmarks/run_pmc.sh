#!/bin/bash
# run_pmc.sh <outdir> <script.py> [script args] -- PASS "<counters>" PASS "<counters>" ...
# One rocprofv3 --pmc pass per counter set, each under its own time limit.  Stops at the first pass that was killed
# by its limit (a hang must not be followed by more GPU work); a pass that merely fails (unknown counter) is skipped.
out=$1; shift
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
export TMPDIR=/tmp
i=0
while [ $# -gt 0 ]; do
  [ "$1" = "PASS" ] || { echo "expected PASS"; exit 2; }
  ctrs=$2; shift 2
  i=$((i+1))
  echo "[pmc pass $i] $ctrs"
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc $ctrs -d "$out/pass$i" -o p -- python3 "${args[@]}" > "$out/pass$i.log" 2>&1
  rc=$?
  echo "[pmc pass $i] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i killed at its limit: stopping"; exit $rc; fi
done
exit 0
