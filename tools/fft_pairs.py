import csv,glob,collections,statistics,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if 'fft2' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
seq=[(r['Kernel_Name'][9:45], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
pairs=collections.OrderedDict()
for i in range(0,len(seq)-1,2):
    key=seq[i][0].split('(')[0]+' + '+seq[i+1][0].split('(')[0]
    pairs.setdefault(key,[]).append((seq[i][1],seq[i+1][1]))
for k,v in pairs.items():
    print('%-70s cols %.1f rows %.1f sum %.1f us (n=%d)'%(k, statistics.median(a for a,b in v), statistics.median(b for a,b in v), statistics.median(a+b for a,b in v), len(v)))
