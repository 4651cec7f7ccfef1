#!/bin/bash
# The pool is closed from outside at the start of the session ("for now"): retry a refused gpurun call every few minutes until
# one is accepted, then stop (the caller looks at the result and drives the next calls by hand).  A refused call costs nothing.
#   tools/gpu_poll.sh <minutes to keep trying> [script to run on the box]
# Every attempt appends one line to gpurun_out/r6_poll.log; the accepted call's output is gpurun_out/r6_poll_call.log.
MIN=${1:-280}; SCRIPT=${2:-tools/gpu_r6_first.sh}
DEADLINE=$(( $(date +%s) + MIN * 60 ))
mkdir -p gpurun_out
while :; do
  NOW=$(date +%s); REM=$(( DEADLINE - NOW ))
  [ $REM -lt 1500 ] && { echo "$(date -u +%T) giving up: $REM s left" >> gpurun_out/r6_poll.log; exit 4; }
  T=$(( REM - 900 )); [ $T -gt 5400 ] && T=5400
  /usr/local/graft/bin/gpurun --timeout $T -- "bash $SCRIPT $(( T - 60 ))" > gpurun_out/r6_poll_call.log 2>&1
  RC=$?
  if grep -q "status=refused" gpurun_out/r6_poll_call.log || [ $RC -eq 3 ]; then
    echo "$(date -u +%T) rc=$RC $(grep -o 'status=[a-z_]*' gpurun_out/r6_poll_call.log | head -1)" >> gpurun_out/r6_poll.log
    sleep 240; continue
  fi
  echo "$(date -u +%T) rc=$RC accepted (timeout $T)" >> gpurun_out/r6_poll.log
  exit $RC
done
