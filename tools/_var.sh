timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py::test_mesh_2d_and_eight_state_rows_at_full_size tests/test_gpu_step_tiles.py tests/test_gpu_tile2.py -q -m gpu -x 2>&1 | tail -15
timeout -k 10 600 python bench.py --steps 20 > gpurun_out/bench2.log 2>&1; tail -c 600 gpurun_out/bench2.log
