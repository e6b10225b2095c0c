python -m pytest tests/test_gemm.py tests/test_modules.py tests/test_dacs.py tests/test_fullsize.py -m gpu -x -q 2>&1 | tail -2
CMDA_HIP_LIB=build/libcmda_hip_timing.so python tools/gemm_phase.py 2>&1 | grep -v amdgpu.ids | head -9
for l in enc enc,T; do echo "== lanes [$l]"; CMDA_BENCH_LANES=$l python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | grep -v "amdgpu\|Graph is empty\|^$" | cut -c80-200; done
python bench.py --workload supervised --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -v "amdgpu" | cut -c80-200
