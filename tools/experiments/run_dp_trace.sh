# kernel trace of the dp64 job on one GPU (16 images per call): timeline of the last evaluate_batch call
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd /tmp
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_dp -o dp -- python $R/bench.py --workload dp64 --dp-images 32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-variants > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(find /tmp/prof_dp -name '*.db' | head -1) 2>&1 | head -70
