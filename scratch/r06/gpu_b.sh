#!/bin/bash
# round 6: (1) the library_user leg alone on the GPU, twice, and once with the caller's own setting; (2) the GPU suite on the
# product library (hook tests on the test build); (3) the whole GPU suite on the test build
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do python bench.py --only-library-user --no-cpu-baseline 2>/dev/null | tail -1; done > gpurun_out/r06_library_user_alone.log
GPU_MAX_HW_QUEUES=16 python bench.py --only-library-user --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r06_library_user_alone.log
GPU_MAX_HW_QUEUES=4 python bench.py --only-library-user --no-cpu-baseline 2>gpurun_out/r06_library_user_q4.err | tail -1 >> gpurun_out/r06_library_user_alone.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06_tests_product.log
EAE_HIP_LIB=test python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06_tests_testlib.log
cat gpurun_out/r06_library_user_alone.log gpurun_out/r06_tests_product.log gpurun_out/r06_tests_testlib.log; grep -i warn gpurun_out/r06_library_user_q4.err | head -5
