cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "presplit or large_tile_gemm or fused_trainer_matches_oracle or config3_dimensions or config2_dimensions or block_resident or test_fit_fused" > gpurun_out/r3_pytest6.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r3_pytest6.log
timeout 300 python tools/gemm_image_ab.py > gpurun_out/r3_gemm_image_ab.txt 2>&1; cat gpurun_out/r3_gemm_image_ab.txt
GSLATES=64 timeout 300 python tools/gemm_image_ab.py > gpurun_out/r3_gemm_image_ab_64.txt 2>&1; cat gpurun_out/r3_gemm_image_ab_64.txt
for r in 1 2; do
  python bench.py --steps 30 --no-cpu-baseline --no-side-pass 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('images   ', d['value'], d['ms_per_step'])"
  python bench.py --steps 30 --no-cpu-baseline --no-side-pass --no-weight-images 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('no images', d['value'], d['ms_per_step'])"
done
