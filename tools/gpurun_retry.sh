#!/bin/bash
# gpurun with retries while every GPU slot of the pod is busy (exit code 3 / "transient"): bash tools/gpurun_retry.sh TIMEOUT 'command'
T=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/gpurun_retry.$$ 2>&1
  if grep -q "status=transient" /tmp/gpurun_retry.$$; then sleep 90; else break; fi
done
cat /tmp/gpurun_retry.$$; rm -f /tmp/gpurun_retry.$$
