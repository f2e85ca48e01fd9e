import csv,re,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
k=list(csv.DictReader(open(f)))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in k]
ev.sort()
calls=[];cur=None
for s,e,n in ev:
    nm=re.sub(r'\(.*','',n).replace('void ','').replace('ixr::','')
    if nm.startswith('k_static_setup'):
        cur=[];calls.append(cur)
    if cur is not None: cur.append((nm,(e-s)/1e3,s,e))
for i in range(5):
    c=calls[i*23+10]
    print(' | '.join('%s %.1f'%(x[0],x[1]) for x in c if x[1]>3), ' span %.1f us'%((c[-1][3]-c[0][2])/1e3))
