"""Per-kernel comparison of two gfx950 assembly listings (hipcc -S --cuda-device-only): instruction streams with the compiler's
local label numbering normalised.  Used to prove that a source change (dead-code removal, comments, adding NEW kernels next to
verified ones) leaves every existing kernel's instructions untouched -- the way to make such a change when no GPU is at hand to
re-run the parity suite.  Exit status 1 if any kernel of the baseline changed or disappeared.
usage: python scripts/isa_equal.py baseline.s candidate.s"""
import re, sys, difflib
def kernels(path):
    out={}; cur=None
    for ln in open(path):
        m=re.match(r'^(_Z\w+):', ln)
        if m: cur=m.group(1); out[cur]=[]; continue
        if cur is None: continue
        t=ln.split(';')[0].rstrip()
        if not t.strip() or re.match(r'\s*\.(loc|file|cfi|ident|section|p2align|type|size|globl|weak|protected|text|amdhsa|end_amdhsa|amdgpu|set)\b', t): continue
        t=re.sub(r'\.LBB\d+_(\d+)', r'.LBB_\1', t.strip()); t=re.sub(r'\.L(tmp|func_end|func_begin)\d+', r'.L\1', t)
        out[cur].append(t)
        if 's_endpgm' in t: cur=None
    return out
a,b=kernels(sys.argv[1]),kernels(sys.argv[2])
diff=[k for k in a if k in b and a[k]!=b[k]]; gone=[k for k in a if k not in b]
print(len(a),'kernels in baseline;', len(a)-len(diff)-len(gone),'identical;', len(diff),'changed;', len(gone),'missing; new:', len(set(b)-set(a)))
for k in diff:
    d=[x for x in difflib.unified_diff(a[k],b[k],lineterm='',n=0) if not x.startswith(('---','+++','@@'))]
    print('  CHANGED', k[:64], len(d),'lines')
sys.exit(1 if diff or gone else 0)
