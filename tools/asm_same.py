#!/usr/bin/env python
"""Are two device assemblies (hipcc --cuda-device-only -S) the same kernel for kernel?  Comments, directives and basic-block
label numbers are ignored.  Used to show that adding an experiment leaves the default kernels' machine code untouched:
  hipcc <flags of fast_lio_amd/_build.py> -x hip fast_lio_amd/csrc/flh_kernels.hip --cuda-device-only -S -o new.s
  python tools/asm_same.py old.s new.s"""
import re, sys
def bodies(path, prefix='_ZN3flh'):
    lines=open(path).read().split('\n'); res={}
    for i,l in enumerate(lines):
        if l.startswith(prefix) and '@_ZN3flh' in l:
            name=l.split(':')[0]; j=i+1; out=[]
            while not lines[j].startswith('.Lfunc_end'):
                s=lines[j].strip(); j+=1
                if not s or s.startswith((';','.')): continue
                out.append(re.sub(r'\.LBB\d+_', '.LBB_', re.sub(r';.*','',s).strip()))
            res[name]=out
    return res
a=bodies(sys.argv[1]); b=bodies(sys.argv[2])
bad=0
for k in a:
    if k not in b: print('missing in new:', k[:100]); bad+=1; continue
    if a[k]!=b[k]:
        import difflib
        d=[x for x in difflib.unified_diff(a[k],b[k],lineterm='',n=0) if not x.startswith(('---','+++','@@'))]
        print('DIFF', k[:100], len(a[k]), len(b[k]), len(d)); bad+=1
        if len(d)<=6: print('   ', d)
print('kernels old', len(a), 'new', len(b), 'differing', bad)
