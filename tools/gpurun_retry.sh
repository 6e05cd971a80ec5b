#!/bin/bash
# developer tool (build container): gpurun with retries while every GPU slot of the pod is busy.  usage: gpurun_retry.sh TIMEOUT 'command'
t=$1; shift
for i in $(seq 1 40); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$t" -- "$@" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"; exit 0
done
echo "gpurun_retry: no slot after 40 attempts"; exit 3
