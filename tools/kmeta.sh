#!/bin/bash
# register / LDS / spill figures of the kernels of one translation unit's object (no GPU needed)
# usage: tools/kmeta.sh nxsearch_amd/csrc/_build/nxs_gpu_scan_bit.o [name filter]
set -e
L=/opt/rocm/lib/llvm/bin
t=$(mktemp -d)
cp "$1" $t/in.o
( cd $t && $L/llvm-objdump --offloading in.o > /dev/null )
for co in $t/*gfx950*; do
$L/llvm-readelf --notes $co | python3 -c "
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r'- \.agpr_count:.*?(?=- \.agpr_count:|\Z)', txt, re.S):
    b=m.group(0)
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)', b) or [None,'?'])[1]
    name=g('name')
    if len(sys.argv)>1 and sys.argv[1] not in name: continue
    print('%-44s vgpr %3s agpr %3s sgpr %3s spill_s %3s spill_v %3s lds %6s scratch %4s' % (name[:44], g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('sgpr_spill_count'), g('vgpr_spill_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
" "${2:-}"
done
rm -rf $t
