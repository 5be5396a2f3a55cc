#!/bin/bash
# round-5 first GPU call: full GPU suite, headline bench lines, scaling decomposition (whole step + per kernel)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=gpurun_out/r05a; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; rc=$?; tail -3 $o/tests.log; [ $rc -ne 0 ] && exit $rc
python bench.py --steps 200 --warmup 20 > $o/bench.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_drv.json 2>> $o/bench.err || exit 1
python tools/scaling.py --out $o/scaling.json 64 128 256 384 512 768 1024 > $o/scaling.txt 2>&1 || { tail -5 $o/scaling.txt; exit 1; }
cat $o/scaling.txt
for n in 128 256 512 1024; do tools/prof.sh r05a/prof$n --batch $n || exit 1; done
python tools/scaling.py --prof 128=$o/prof128 256=$o/prof256 512=$o/prof512 1024=$o/prof1024 --out $o/kernel_scaling.json > $o/kernel_scaling.txt 2>&1
cat $o/kernel_scaling.txt
