#!/bin/bash
# scripts/sweep_env.sh VAR "v1 v2 ..." <command ...>: run the command once per value of the environment variable
V=$1; VALS=$2; shift 2
for t in $VALS; do
  echo "== $V=$t"
  env $V=$t timeout -k 10 300 "$@" || exit 1
done
