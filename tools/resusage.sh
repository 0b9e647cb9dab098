#!/bin/bash
# print VGPR / LDS / scratch usage of every kernel in a .hip file: tools/resusage.sh file.hip [extra hipcc flags]
set -e
src=$(realpath "$1"); shift
tmp=$(mktemp -d); cd "$tmp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c "$src" -save-temps -o x.o "$@" 2>/dev/null
python3 - <<'PY'
import re,glob,subprocess
s=open(glob.glob('*gfx950.s')[0]).read()
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size', s, re.S):
    blk=m.group(0)
    g=lambda k: re.search(r'\.%s:\s+(\S+)'%k, blk).group(1)
    name=subprocess.run(['c++filt',g('name')],capture_output=True,text=True).stdout.strip()
    print('%-110s vgpr %4s agpr %4s sgpr %4s lds %7s scratch %5s'%(name[:110],g('vgpr_count'),g('agpr_count'),g('sgpr_count'),g('group_segment_fixed_size'),g('private_segment_fixed_size')))
PY
rm -rf "$tmp"
