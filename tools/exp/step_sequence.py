import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'][:60] for r in rows]
# find last adam kernel index and print the step before it
idx=[i for i,n in enumerate(names) if 'adam_kernel' in n]
a,b=idx[-2],idx[-1]
for i in range(a,b+1):
    r=rows[i]; print(i-a, names[i], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
