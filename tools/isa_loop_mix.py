import re,collections,sys
MINM=int(__import__("os").environ.get("MINM","200"))
def analyze(path, kernel_sub, quiet=False):
    txt=open(path).read().split('\n')
    # find function start/end
    start=None
    for i,l in enumerate(txt):
        if l.startswith('_ZN') and kernel_sub in l and ':' in l and start is None:
            start=i
        if start is not None and l.strip().startswith('.end_amdhsa_kernel') : pass
        if start is not None and l.startswith('.Lfunc_end'):
            end=i; break
    lines=txt[start:end]
    labels={}
    for i,l in enumerate(lines):
        m=re.match(r'^(\.LBB\d+_\d+):',l)
        if m: labels[m.group(1)]=i
    mf=[i for i,l in enumerate(lines) if 'v_mfma' in l]
    tr=[i for i,l in enumerate(lines) if 'v_exp_f32' in l]
    MINT=int(__import__('os').environ.get('MINT','0'))
    best=None
    for i,l in enumerate(lines):
        m=re.search(r's_cbranch\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)',l)
        if m:
            t=m.group(1) or m.group(2)
            if labels.get(t,1e9)<i:
                n=sum(1 for j in mf if labels[t]<=j<=i)
                nt_=sum(1 for j in tr if labels[t]<=j<=i)
                if n>=MINM and nt_>=MINT and (best is None or (i-labels[t])<best[1]-best[0]):
                    best=(labels[t],i,n)
    body=lines[best[0]:best[1]+1]
    c=collections.Counter()
    for l in body:
        l=l.strip()
        if not l or l.startswith(('.',';','/')) or l.endswith(':'): continue
        c[l.split()[0]]+=1
    cls=collections.Counter()
    cyc=0
    for op,n in c.items():
        if op.startswith('v_mfma'): k='mfma'
        elif op.startswith(('v_exp','v_rcp','v_sqrt','v_rsq','v_log')): k='trans'
        elif op.startswith('v_pk'): k='vpk'
        elif op.startswith('v_'): k='valu'
        elif op.startswith('ds_'): k='lds'
        elif op.startswith(('global_','buffer_','scratch_','flat_')): k='vmem'
        elif op.startswith('s_waitcnt'): k='wait'
        elif op.startswith('s_barrier'): k='barrier'
        elif op.startswith('s_'): k='salu'
        else: k='other'
        cls[k]+=n
    print(kernel_sub, 'loop lines',best, dict(cls))
    if not quiet:
        for op,n in c.most_common(45): print('   ',n,op)
if __name__=='__main__':
    analyze(sys.argv[1], sys.argv[2], len(sys.argv)>3)
