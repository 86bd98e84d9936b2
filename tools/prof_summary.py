import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel                                                                                      calls/step   us/step   avg us    %")
for r in rows[:28]:
    n = r['Name'].replace('void vmmt::', '').replace('unsigned short', 'bf16')
    print("%-90s %8.1f %9.1f %8.2f %5.1f" % (n[:90], int(r['Calls'])/nsteps, float(r['TotalDurationNs'])/1e3/nsteps, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
print("total GPU kernel time per step: %.3f ms" % (tot/1e6/nsteps))
