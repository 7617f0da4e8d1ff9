#!/bin/bash
# round-2 run C: whole GPU suite (no -x) + smoke
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.txt
( time timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -rA ) > gpurun_out/pytest_all.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_all.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_all.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
