import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
want=sys.argv[2:] 
for r in rows:
    m=re.search(r"\bk_\w+",r['Name']); n=m.group(0) if m else r['Name'][:30]
    if not want or n in want: print(f"{n:24s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us")
