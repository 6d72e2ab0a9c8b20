#!/bin/bash
# tools/disasm.sh [OBJECT] -> /tmp/bnn_disasm/k.s: symbolised gfx950 disassembly of the kernels object (demangled),
# and /tmp/bnn_disasm/k.notes: the code object's metadata (register counts, LDS, spills per kernel)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OBJ=${1:-$R/bnn-pynq_amd/build/kernels.o}
T=/tmp/bnn_disasm; rm -rf $T; mkdir -p $T; cp $OBJ $T/k.o; cd $T
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading k.o > /dev/null
CO=$(ls | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d --symbolize-operands --no-show-raw-insn $CO | c++filt > k.s
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $CO | c++filt > k.notes
echo $T/k.s
