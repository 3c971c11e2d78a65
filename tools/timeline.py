import csv,glob,sys
f=glob.glob(sys.argv[1],recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('k_primary')][-1]
t0=int(rows[idx]['Start_Timestamp'])
for r in rows[idx:]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} -> {(int(r['End_Timestamp'])-t0)/1e3:9.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size'))} q{r.get('Queue_Id','?')} {r['Kernel_Name'][:36]}")
