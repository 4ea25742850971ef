"""Join the kernels of the last call of a `rocprofv3 --kernel-trace` run of tools/cascade_probe.py (BSX_DEBUG=1) with the
library's per-level debug lines, in launch order:  python tools/levels_join.py <dir>   (expects <dir>/p63_kernel_trace.csv and
<dir>.log; profiles/r03_levels.md is its output).  One stream only (BSX_CUBE_STREAMS=1), or the order is not the chains'."""
import csv,re,sys
d=sys.argv[1]
rows=list(csv.DictReader(open(d+'/p63_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_publish' in r['Kernel_Name']]
seg=rows[idx[-2]+1:idx[-1]+1]
ks=[]
for r in seg:
    n=r['Kernel_Name']; dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if 'k_attract_pool' in n: ks.append(('top' if 'true, false>' in n else 'low', dur))
cmp_total=sum((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in seg if 'compact' in r['Kernel_Name'])
span=(int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e3
alll=open(d+'.log').read().split('[bsx] attract:')
last=[l for l in alll[-1].split('\n') if l.startswith('[bsx] cube')]
pat=re.compile(r'depth (\d+)( \(top\)| \(per parent\))?, (\d+) digits here \((\d+) relevant\), (\d+) classes, (\d+) near')
lv=[]
for l in last:
    m=pat.search(l); lv.append((int(m.group(1)),(m.group(2) or '').strip(),int(m.group(3)),int(m.group(4)),int(m.group(5)),int(m.group(6))))
ki=0; li=0
tot={}
print('span us %.0f, compact total %.0f'%(span,cmp_total))
while li<len(lv):
    dd,kind,dig,rel,cl,near=lv[li]
    chain=[lv[li]]; li+=1
    while li<len(lv) and lv[li][1]!='(top)': chain.append(lv[li]); li+=1
    T=dd
    kk=ks[ki:ki+T]; ki+=T
    s=[]
    for j in range(T):
        if j<len(chain):
            c=chain[j]; s.append('d%d%s 2^%.1f %.0fus'%(c[0],'P' if 'parent' in c[1] else '', __import__('math').log2(max(c[4],1)),kk[j][1]))
            key='top' if j==0 else ('leaf' if 'parent' in c[1] else 'child')
        else:
            s.append('empty %.0fus'%kk[j][1]); key='empty'
        tot[key]=tot.get(key,0)+kk[j][1]
    print(' | '.join(s))
print(tot)
