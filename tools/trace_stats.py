import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].replace('void (anonymous namespace)::','')[:48]
    k=(n,r['Grid_Size_X'],r['Grid_Size_Y'])
    agg[k][0]+=1; agg[k][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for k,v in sorted(agg.items()):
    if v[0]>=20: print(f'{k[0]:50s} {k[1]:>8s} {k[2]:>6s} n={v[0]:4d} avg={v[1]/v[0]:8.1f} us')
