#!/bin/bash
# Run a list of bench.py configurations on the GPU box in one gpurun call:
#   tools/bench_many.sh <outdir> <file with one "label | bench.py args" per line>
# One JSON line per configuration in <outdir>/<label>.json (stderr in .err).  Stops at the first run that is killed by its
# timeout (a GPU command that hung tells something: no further GPU step in the same call).
out="$1"; list="$2"
mkdir -p "$out"
while IFS='|' read -r label args; do
  label=$(echo "$label" | xargs); [ -z "$label" ] && continue
  case "$label" in \#*) continue;; esac
  echo "== $label: bench.py $args"
  timeout -k 10 300 python3 bench.py $args > "$out/$label.json" 2> "$out/$label.err"
  rc=$?
  echo "   rc=$rc $(head -c 300 "$out/$label.json")"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done < "$list"
