cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dec; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pd1 -o d -- python $R/tools/bench_decode.py --reps 3 > $O/p1.log 2>&1
python $R/tools/rocpd_by_grid.py $(find /tmp/pd1 -name "*.db" | head -1) "" > $O/shapes1.txt 2>&1
python $R/tools/rocpd_rooflines.py $(find /tmp/pd1 -name "*.db" | head -1) > $O/roof1.txt 2>&1
