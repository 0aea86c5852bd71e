#!/bin/bash
# repeat the bring-up check a few times, stop at the first failure/timeout
for i in 1 2 3 4 5 6; do
  timeout -k 10 60 python tools/queue_check.py 2>&1 | grep -v amdgpu.ids | tail -5 || { echo "RUN $i FAILED rc=$?"; exit 1; }
  echo "--- run $i ok"
done
