#!/bin/bash
# Every GPU test file in a process of its own (order independence: each must pass without the others having run first).
# Run on the GPU box from the repository root; stops at the first file that fails or times out.
cd "$(dirname "$0")/.." || exit 1
for f in tests/test_*.py; do
  grep -q "mark.gpu" "$f" || continue
  timeout -k 10 400 python3 -u -m pytest "$f" -m gpu -x -q --timeout 300 --timeout-method thread > /tmp/one.txt 2>&1
  rc=$?
  echo "$f rc=$rc $(tail -1 /tmp/one.txt)"
  [ $rc -eq 0 ] || [ $rc -eq 5 ] || { tail -30 /tmp/one.txt; exit 1; }
done
