#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3g
mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_models_gpu.py tests/test_properties.py -m gpu -q -k "cin or CIN or xdeepfm or XDeepFM or zoo" > $out/pytest_cin.log 2>&1
echo "pytest rc=$?" >> $out/pytest_cin.log
tail -6 $out/pytest_cin.log
