"""Load / wait / LDS-store / barrier / MFMA sequence of every kernel in a device assembly file (tools/kres.sh leaves /tmp/kres_<file>.s):
python tools/isa_loadwait.py /tmp/kres_X.s [name filter]   -- repeated "L.. W(0) s" groups = dependent memory round trips in a staging loop."""
import itertools, re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\S+):.*\n', txt, re.M):
    name = m.group(1)
    if flt not in name:
        continue
    i = m.end(); j = txt.find('.Lfunc_end', i)      # (a kernel may hold several s_endpgm: early returns)
    if j < 0:
        continue
    seq = []
    for l in txt[i:j].split('\n'):
        l = l.strip()
        if l.startswith(('global_load', 'buffer_load')): seq.append('L')
        elif l.startswith('s_waitcnt') and 'vmcnt' in l: seq.append('W%s' % re.search(r'vmcnt\((\d+)\)', l).group(1))
        elif l.startswith(('ds_write', 'ds_store')): seq.append('s')
        elif l.startswith('s_barrier'): seq.append('|')
        elif l.startswith('v_mfma'): seq.append('M')
        elif l.startswith(('s_cbranch', 's_branch')): seq.append('^')
    out = []
    for k, g in itertools.groupby(seq):
        n = len(list(g)); out.append(k if n == 1 else f"{k}x{n}")
    print(name[:90]); print('   ' + ' '.join(out)[:2400]); print()
