mkdir -p gpurun_out/r3p
timeout -k 10 1100 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3p/tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3p/tests.log
exit $rc
