python -m pytest tests/test_dense_gpu.py tests/test_parity_mode_gpu.py tests/test_stages_gpu.py tests/test_fp8_gpu.py -x -q -k "not full_depth" 2>&1 | tail -2
for i in 1 2 3; do
  echo -n "all on 320: "; N=10 python tools/bench_encoder_modes.py 2>&1 | tail -2 | tr '\n' ' '; echo
  echo -n "fp32 only:  "; IVLM_320_F32ONLY=1 N=10 python tools/bench_encoder_modes.py 2>&1 | tail -2 | tr '\n' ' '; echo
done
