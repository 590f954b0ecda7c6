# Round profile: GPU tests, the driver-style bench line, rocprofv3 kernel trace of the headline loop (graphs ON), PMC traffic
# passes (graphs off: counter collection crashes on replayed graphs), two-stream timeline.  Results under gpurun_out/r6/.
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
cd $R
timeout 1500 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
timeout 300 python tools/bench_lift.py > $O/lift.json 2> $O/lift.err
cd /tmp
# the headline loop alone, HIP graphs ON (the regime `value` is measured in)
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-roofline > $O/bench_kt.json 2> $O/kt.err
DB=$(find /tmp/prof_kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB 12 > $O/kernel_stats_graphs.txt 2>&1
python $R/tools/rocpd_rooflines.py $DB >> $O/kernel_stats_graphs.txt 2>&1
python $R/tools/rocpd_timeline.py $DB > $O/timeline.txt 2>&1
IVLM_NO_ADVERSARIAL=1 IVLM_NO_GRAPHS=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-roofline > /dev/null 2> $O/pmc_f.err
IVLM_NO_ADVERSARIAL=1 IVLM_NO_GRAPHS=1 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-roofline > /dev/null 2> $O/pmc_w.err
python $R/tools/rocpd_pmc.py $(find /tmp/prof_f -name '*.db' | head -1) $(find /tmp/prof_w -name '*.db' | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
timeout 600 python $R/bench.py --model 13b --steps 5 --warmup 2 --no-variants > $O/bench_13b.json 2> $O/bench_13b.err
tail -2 $O/kernel_stats_graphs.txt; head -c 600 $O/bench.json
