set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --detector pixel --no-cpu-baseline > $O/bench_pixel.json 2> $O/bench_pixel.err
python bench.py --task yaw --no-cpu-baseline > $O/bench_yaw.json 2> $O/bench_yaw.err
python tools/step_sweep.py $O/sweep_timing.json > $O/sweep.log 2>&1
python tools/step_sweep.py $O/sweep_timing_pixel.json pixel > $O/sweep_pixel.log 2>&1
for N in 4096 65536 524288 4194304 16777216; do
  L=10; if [ $N -ge 4194304 ]; then L=4; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc/fetch_$N -- python3 tools/pmc_step.py $N $L > $O/pmc_fetch_$N.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc/write_$N -- python3 tools/pmc_step.py $N $L > $O/pmc_write_$N.log 2>&1
  echo pmc $N done
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU --output-format csv -d $O/pmc/sq_a -- python3 tools/pmc_step.py 65536 10 > $O/pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc/sq_b -- python3 tools/pmc_step.py 65536 10 > $O/pmc_sq_b.log 2>&1
echo sq done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hover -- python3 bench.py --steps 256 --warmup 32 --repeats 2 --no-cpu-baseline --no-split-extra > $O/prof_hover.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_yaw -- python3 bench.py --task yaw --steps 256 --warmup 32 --repeats 2 --no-cpu-baseline > $O/prof_yaw.log 2>&1
echo prof done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
du -sh $O
