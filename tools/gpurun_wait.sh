#!/bin/bash
# tools/gpurun_wait.sh TIMEOUT 'COMMAND': gpurun, retried every 2 minutes while no GPU slot / box is free (exit code 3:
# nothing was charged, nothing ran); any other outcome is returned as it is
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
