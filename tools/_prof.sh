cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dec; mkdir -p $O
cd /tmp
for v in 1; do
IVLM_DECODE_PACKED=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pd$v -o d -- python $R/tools/bench_decode.py --reps 3 > $O/p$v.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/pd$v -name '*.db' | head -1) 8 > $O/stats$v.txt 2>&1
python $R/tools/rocpd_by_grid.py $(find /tmp/pd$v -name "*.db" | head -1) gemv > $O/shapes$v.txt 2>&1
done
