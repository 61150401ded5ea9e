#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/s12
mkdir -p $O
export LIDIFF_PARITY_LOG=$PWD/$O/parity_errors.jsonl
rm -f $LIDIFF_PARITY_LOG
timeout 1700 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -15 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
