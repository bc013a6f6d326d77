#!/bin/bash
# full -m gpu suite, then the trainer profile (its digest covers every kernel source), then the complete default bench line
set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r04_final_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_final_tests.log
bash tools/profile_train.sh r04 > gpurun_out/r04_train_profile.log 2>&1; echo "profile_train rc=$?"
