set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2j; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hover -- python3 bench.py --steps 256 --warmup 32 --repeats 2 --no-cpu-baseline --no-split-extra > $O/prof_hover.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 128 --warmup 32 --repeats 1 --no-cpu-baseline --no-split-extra --kernel-launches 20 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 128 --warmup 32 --repeats 1 --no-cpu-baseline --no-split-extra --kernel-launches 20 > $O/pmc_write.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
du -sh $O
