# MFMA utilisation per kernel of the headline loop from SQ counters (own passes: --pmc with --kernel-trace only; graphs off - counter
# collection crashes on replayed graphs).  Results: gpurun_out/r6g/pmc_mfma_util.txt
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6g; mkdir -p $O
cd /tmp
IVLM_NO_ADVERSARIAL=1 IVLM_NO_GRAPHS=1 timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA -d /tmp/prof_sq -o sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-roofline > /dev/null 2> $O/pmc_sq.err
IVLM_NO_ADVERSARIAL=1 IVLM_NO_GRAPHS=1 timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d /tmp/prof_gr -o gr -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-roofline > /dev/null 2> $O/pmc_gr.err
python $R/tools/rocpd_mfma.py $(find /tmp/prof_sq -name '*.db' | head -1) $(find /tmp/prof_gr -name '*.db' | head -1) > $O/pmc_mfma_util.txt 2>&1
cat $O/pmc_mfma_util.txt; tail -3 $O/pmc_sq.err
