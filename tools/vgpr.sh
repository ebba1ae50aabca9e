#!/bin/bash
# register / LDS census of the kernels of one .hip file, offline (no GPU):  tools/vgpr.sh visiondepth3d_amd/csrc/vd3d_finish.hip [extra hipcc flags]
F=$(realpath $1); shift
D=$(mktemp -d); cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -Wno-unused-function -I$(dirname $F) "$@" -c $F -o x.o --save-temps 2>/dev/null
python3 - <<'PY'
import re,glob
s=open(glob.glob('*gfx950.s')[0]).read()
for m in re.finditer(r'\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count: (\d+)', s, re.S):
    print(f"vgpr {m.group(4):>4} sgpr {m.group(3):>4} lds {m.group(1):>6} spill {m.group(5):>3}  {m.group(2)[:90]}")
PY
cp *gfx950.s /tmp/last_kernel.s
rm -rf $D
