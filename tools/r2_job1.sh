set -e
mkdir -p gpurun_out/r2a
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r2a/counters.txt 2>&1 || true
cd $GRAFT_REPO_ROOT
python3 tools/stage_times.py lfcW1A1 10000 8192 16384 131072 > gpurun_out/r2a/stage_lfcW1A1.txt 2>&1
python3 tools/stage_times.py lfcW1A2 10000 131072 > gpurun_out/r2a/stage_lfcW1A2.txt 2>&1
python3 tools/stage_times.py cnvW1A2 131072 > gpurun_out/r2a/stage_cnvW1A2.txt 2>&1
python3 tools/stage_times.py cnvW2A2 131072 > gpurun_out/r2a/stage_cnvW2A2.txt 2>&1
python3 tools/stage_times.py cnvW1A1 10000 131072 > gpurun_out/r2a/stage_cnvW1A1.txt 2>&1
cat gpurun_out/r2a/stage_*.txt
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/sq1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/r2a/sq1.json 2>$GRAFT_REPO_ROOT/gpurun_out/r2a/sq1.err
echo sq1 done
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/r2a/sq2.json 2>$GRAFT_REPO_ROOT/gpurun_out/r2a/sq2.err
echo sq2 done
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/grbm -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/r2a/grbm.json 2>$GRAFT_REPO_ROOT/gpurun_out/r2a/grbm.err
echo grbm done
