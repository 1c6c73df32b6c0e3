"""Per-basic-block instruction census of one kernel in a gfx950 .s file (scratch / global / LDS / barrier / MFMA):
where the spills of a kernel sit relative to its loops.   python tools/asm_blocks.py FILE.s KERNEL_SUBSTRING"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
names = re.findall(r'^(_Z\w+):', s, flags=re.M)
name = [n for n in names if key in n][int(sys.argv[3]) if len(sys.argv) > 3 else 0]
i = s.index(name + ':')
body = s[i:s.index('s_endpgm', i)]
blk, stats, order = 'entry', {}, []
for ln in body.split('\n'):
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m:
        blk = m.group(1)
    t = ln.strip()
    if blk not in stats:
        stats[blk] = dict(n=0, scr_ld=0, scr_st=0, gl=0, gs=0, ds=0, bar=0, mfma=0, f64=0, br='')
        order.append(blk)
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
        continue
    st = stats[blk]
    st['n'] += 1
    if t.startswith('scratch_load'): st['scr_ld'] += 1
    if t.startswith('scratch_store'): st['scr_st'] += 1
    if t.startswith('global_load'): st['gl'] += 1
    if t.startswith('global_store') or t.startswith('global_atomic'): st['gs'] += 1
    if t.startswith('ds_'): st['ds'] += 1
    if t.startswith('s_barrier'): st['bar'] += 1
    if 'mfma' in t: st['mfma'] += 1
    if '_f64' in t: st['f64'] += 1
    if t.startswith('s_cbranch') or t.startswith('s_branch'): st['br'] += t.split()[-1] + ' '
print(name, len(body.split('\n')), 'lines')
for b in order:
    st = stats[b]
    print('%-12s n=%4d f64=%4d scrL=%3d scrS=%3d gl=%3d gs=%2d ds=%3d bar=%d mfma=%2d -> %s' % (
        b, st['n'], st['f64'], st['scr_ld'], st['scr_st'], st['gl'], st['gs'], st['ds'], st['bar'], st['mfma'], st['br']))
