set -e
mkdir -p gpurun_out/r2j
cd /tmp && export TMPDIR=/tmp
export BNN_MI355X_LFC_BLOCK_MAX=200000 BATCHES=131072 BNN_MI355X_LFC_BLOCK=s
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2j/sq1 -- python3 $GRAFT_REPO_ROOT/tools/batch_sweep.py lfcW1A1 > $GRAFT_REPO_ROOT/gpurun_out/r2j/sq1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2j/sq2 -- python3 $GRAFT_REPO_ROOT/tools/batch_sweep.py lfcW1A1 > $GRAFT_REPO_ROOT/gpurun_out/r2j/sq2.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2j/grbm -- python3 $GRAFT_REPO_ROOT/tools/batch_sweep.py lfcW1A1 > $GRAFT_REPO_ROOT/gpurun_out/r2j/grbm.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/sq_summary.py gpurun_out/r2j/sq1 gpurun_out/r2j/sq2 gpurun_out/r2j/grbm > gpurun_out/r2j/sq_summary.json
python3 -c "import json; d=json.load(open('gpurun_out/r2j/sq_summary.json')); print(json.dumps(d['k_lfc_block_s'], indent=1))"
