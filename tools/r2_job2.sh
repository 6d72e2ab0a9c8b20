set -e
mkdir -p gpurun_out/r2b
for sp in 0 1; do
  echo "== BNN_MI355X_SPREAD=$sp"
  BNN_MI355X_SPREAD=$sp python3 tools/stage_times.py lfcW1A1 2048 4096 6000 8192 10000 16384 32768 65536 131072
  BNN_MI355X_SPREAD=$sp python3 tools/stage_times.py cnvW1A1 512 1024 2048 4096 10000 131072
done > gpurun_out/r2b/spread.txt 2>&1
cat gpurun_out/r2b/spread.txt | grep -v amdgpu.ids
python3 -c "
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'bnn-pynq_amd')
import torch, bnn, oracle_lib as ol, os
clf = bnn.LfcClassifier(bnn.NETWORK_LFCW1A1, 'mnist', bnn.RUNTIME_SW)
print(clf.classify_mnist(os.path.join(ol.GOLDEN, '3.image-idx3-ubyte')), clf.usecPerImage)
print(clf.classify_mnist(os.path.join(ol.GOLDEN, '3.image-idx3-ubyte')), clf.usecPerImage)
"
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15
