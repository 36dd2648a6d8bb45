#!/bin/bash
# Register / scratch / LDS metadata of every kernel in ONE device object or library (a .o of a .hip file, or a .so), and the
# disassembly of the whole code object under /tmp/obj_regs/all.s when a name pattern is given.
# usage: tools/obj_regs.sh file.o|file.so [name-pattern]
f=$1; pat=${2:-}
tmp=$(mktemp -d); cp "$f" $tmp/in.bin
(cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading in.bin > /dev/null 2>&1)
for co in $tmp/in.bin.*gfx950*; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" 2>/dev/null |
    grep -E "^ +\.name:|\.vgpr_count|vgpr_spill|private_segment_fixed|\.sgpr_count:|group_segment_fixed|sgpr_spill" | paste - - - - - - - |
    sed -E 's/[ \t]+/ /g; s/\.private_segment_fixed_size/scratch/; s/\.group_segment_fixed_size/lds/; s/\.vgpr_spill_count/vspill/; s/\.sgpr_spill_count/sspill/' | { [ -n "$pat" ] && grep -E "$pat" || cat; }
  if [ -n "$pat" ]; then
    mkdir -p /tmp/obj_regs
    /opt/rocm/lib/llvm/bin/llvm-objdump -d "$co" > /tmp/obj_regs/all.s
  fi
done
rm -rf $tmp
