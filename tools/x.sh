python -m pytest tests/test_train_gpu.py tests/test_train_fullsize_gpu.py tests/test_stage2_gpu.py -m gpu -x -q > gpurun_out/r04ab_tests.log 2>&1 || true
tail -5 gpurun_out/r04ab_tests.log
python3 tools/train_ab.py "" head 2>&1 | grep -v amdgpu.ids
