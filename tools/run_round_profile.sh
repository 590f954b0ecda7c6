set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
timeout 300 python tools/bench_lift.py > $O/lift.json 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants > $O/bench_kt.json 2> $O/kt.err
python $R/tools/rocpd_stats.py $(find /tmp/prof_kt -name '*.db' | head -1) 7 > $O/kernel_stats.txt 2>&1
IVLM_NO_ADVERSARIAL=1 IVLM_NO_GRAPHS=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > /dev/null 2> $O/pmc_f.err
IVLM_NO_ADVERSARIAL=1 IVLM_NO_GRAPHS=1 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > /dev/null 2> $O/pmc_w.err
python $R/tools/rocpd_pmc.py $(find /tmp/prof_f -name '*.db' | head -1) $(find /tmp/prof_w -name '*.db' | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
tail -3 $O/pytest.log; cat $O/bench.json; cat $O/lift.json
