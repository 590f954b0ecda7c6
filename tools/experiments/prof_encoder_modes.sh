export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/enc; mkdir -p $O
cd $R
MODES=default,f16,f16q N=10 python tools/bench_encoder_modes.py 2>&1 | tail -4
cd /tmp
for m in default f16q; do
  MODES=$m N=3 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o kt -- python $R/tools/bench_encoder_modes.py > /dev/null 2> $O/$m.err
  python $R/tools/rocpd_stats.py $(find /tmp/prof_$m -name '*.db' | head -1) 5 > $O/stats_$m.txt 2>&1
  head -24 $O/stats_$m.txt
done
