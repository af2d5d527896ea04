#!/bin/bash
# round 5, last GPU session on the committed tree: smoke(), the whole GPU suite
export TMPDIR=/tmp
T=${R05TAG:-r05C}
mkdir -p gpurun_out/$T
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$T/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/$T/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$T/gputest.txt 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/$T/gputest.txt
