python -m pytest tests/test_render_gpu.py tests/test_fullframe_gpu.py tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/perf_ab.py --rounds 5 --frames 10 --configs "lib=;lib=head" 2>&1 | grep median
timeout -k 10 200 python3 tools/cu_steal_probe.py --hog-cus 16 --hog-us 100 300 600 2>&1 | tail -1
