#!/bin/bash
# Run the GPU steps given as arguments ("name::command" each) one after another on the gpurun box.  A step that fails with an
# ordinary exit code (a red test) is recorded and the next one still runs; a step that TIMES OUT or is KILLED (124, 137, any signal)
# ends the call — no further GPU step is started after a hang.  Each step's output goes to gpurun_out/<name>.log.
mkdir -p gpurun_out
rc_all=0
for step in "$@"; do
  name="${step%%::*}"; cmd="${step#*::}"
  echo "=== $name: $cmd" | tee -a gpurun_out/steps.log
  timeout -k 10 ${STEP_TIMEOUT:-900} bash -o pipefail -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/steps.log
  tail -n ${STEP_TAIL:-6} "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "step $name timed out / was killed: stopping" | tee -a gpurun_out/steps.log; exit $rc; fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
