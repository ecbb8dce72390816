#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh r03
# kernel statistics of the four benches, the PMC traffic passes (FETCH_SIZE and WRITE_SIZE separately) and the JSON lines.
set -o pipefail
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[collect] DCGAN kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/dcgan -o p -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-calib --no-secondary > $O/dcgan.log 2>&1 || exit 1
echo "[collect] DCGAN pmc fetch";    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calib --no-secondary > $O/pmc_fetch.log 2>&1 || exit 1
echo "[collect] DCGAN pmc write";    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calib --no-secondary > $O/pmc_write.log 2>&1 || exit 1
echo "[collect] countergan";         rocprofv3 --kernel-trace --stats --output-format csv -d $O/countergan -o p -- python3 $R/scripts/bench_countergan.py --steps 5 --warmup 2 --no-cpu-baseline > $O/countergan.log 2>&1 || exit 1
echo "[collect] wgan";               rocprofv3 --kernel-trace --stats --output-format csv -d $O/wgan -o p -- python3 $R/scripts/bench_wgan.py --steps 10 --warmup 2 --no-cpu-baseline > $O/wgan.log 2>&1 || exit 1
echo "[collect] countergan pmc";     rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_countergan -o p -- python3 $R/scripts/bench_countergan.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_countergan.log 2>&1 || exit 1
                                     rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_countergan -o p -- python3 $R/scripts/bench_countergan.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_countergan.log 2>&1 || exit 1
echo "[collect] wgan pmc";           rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_wgan -o p -- python3 $R/scripts/bench_wgan.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_wgan.log 2>&1 || exit 1
                                     rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_wgan -o p -- python3 $R/scripts/bench_wgan.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write_wgan.log 2>&1 || exit 1
echo "[collect] house";              rocprofv3 --kernel-trace --stats --output-format csv -d $O/house -o p -- python3 $R/scripts/bench_house.py --steps 50 --warmup 10 --no-cpu-baseline > $O/house.log 2>&1 || exit 1
cd $R
echo "[collect] JSON lines"
python3 bench.py > $O/bench_line.json 2> $O/bench_line.err || exit 1
python3 bench.py --force-dp --no-cpu-baseline --no-secondary > $O/bench_line_dp1.json 2> $O/bench_line_dp1.err || exit 1
python3 scripts/bench_countergan.py > $O/countergan_line.json 2> $O/countergan_line.err || exit 1
python3 scripts/bench_wgan.py > $O/wgan_line.json 2> $O/wgan_line.err || exit 1
python3 scripts/bench_house.py > $O/house_line.json 2> $O/house_line.err || exit 1
echo "[collect] done"
