#!/bin/bash
# Round 6: the drop-in on the reference's own class.  The reference checkout is shipped to the GPU box as untracked, git-ignored
# scratch (_ref_scratch/, removed right after the call; never committed) and named by MOONDREAM_REFERENCE.
R=$GRAFT_REPO_ROOT
export MOONDREAM_REFERENCE=$R/_ref_scratch
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests/test_dropin_gpu.py -m gpu -x -q -rs 2>&1 | tail -25 > gpurun_out/r06_dropin_pytest.txt
cat gpurun_out/r06_dropin_pytest.txt
python tools/dropin_report.py > gpurun_out/r06_dropin_report.json 2> gpurun_out/r06_dropin_report.err
tail -3 gpurun_out/r06_dropin_report.err; cat gpurun_out/r06_dropin_report.json
