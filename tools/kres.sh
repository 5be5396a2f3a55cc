#!/bin/bash
# usage: tools/kres.sh FILE.hip [extra flags]  -> per-kernel VGPRs / SGPRs / LDS bytes / scratch bytes / spills (device ISA metadata)
src=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$(ls -d $root/*_amd/csrc)
out=/tmp/kres_$(basename $src .hip).s
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -I $root/include -I $csrc "$@" --cuda-device-only -S $csrc/$src -o $out || exit 1
python3 - $out <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:", txt, re.S):
    blk = m.group(0)
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    print(f"{g('name')[:70]:70s} vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>3} sgpr {g('sgpr_count'):>4} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>5} spill {g('vgpr_spill_count')}")
PY
