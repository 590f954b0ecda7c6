# Round-5 measurement batch (one gpurun call): per-shape GEMM ceilings (product + ablation libraries), the prefill tile x split-K
# sweep, the x-staging bound of the decode step, stage times alone.  Results under gpurun_out/r5x/.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5x; mkdir -p $O
cd $R
IVLM_SPLITK_FUSED=0 python tools/bench_gemm_ceilings.py --json $O/ceil_product.json > $O/ceil_product.log 2>&1
python tools/bench_gemm_ceilings.py --json $O/ceil_product_fused_splitk.json > $O/ceil_product_fused_splitk.log 2>&1
for t in noepi mfma_lds mfma_dma mfma_only dma_lds dma_only; do
  [ -f tools/_bin/libivlm_$t.so ] && IVLM_SPLITK_FUSED=0 IVLM_LIB_PATH=$R/tools/_bin/libivlm_$t.so timeout 600 python tools/bench_gemm_ceilings.py --json $O/ceil_$t.json > $O/ceil_$t.log 2>&1
done
python tools/gemm_ceilings_table.py $O > $O/gemm_ceilings.txt 2>&1
timeout 900 python tools/bench_prefill_gemm.py > $O/prefill_sweep.txt 2>&1
timeout 600 python tools/bench_decode.py > $O/decode_product.txt 2>&1
IVLM_SPLITK_FUSED=0 IVLM_LIB_PATH=$R/tools/_bin/libivlm_xstage.so timeout 600 python tools/bench_decode.py > $O/decode_xstage.txt 2>&1
timeout 600 python tools/bench_decode.py >> $O/decode_product.txt 2>&1
IVLM_SPLITK_FUSED=0 IVLM_LIB_PATH=$R/tools/_bin/libivlm_xstage.so timeout 600 python tools/bench_decode.py >> $O/decode_xstage.txt 2>&1
timeout 900 python tools/bench_stages.py > $O/stages.txt 2>&1
IVLM_SPLITK_FUSED=0 timeout 900 python tools/bench_stages.py > $O/stages_unfused_splitk.txt 2>&1
timeout 900 python -m pytest tests/test_dense_gpu.py tests/test_stages_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest_subset.txt
cat $O/gemm_ceilings.txt; cat $O/ceil_product_fused_splitk.log; cat $O/prefill_sweep.txt; grep ms/token $O/decode_product.txt $O/decode_xstage.txt; tail -6 $O/stages.txt; tail -6 $O/stages_unfused_splitk.txt; cat $O/pytest_subset.txt
