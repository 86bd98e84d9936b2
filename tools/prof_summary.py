import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel                                                                                      calls/step   us/step   avg us    %")
for r in rows[:28]:
    n = r['Name'].replace('void vmmt::', '').replace('unsigned short', 'bf16')
    print("%-90s %8.1f %9.1f %8.2f %5.1f" % (n[:90], int(r['Calls'])/nsteps, float(r['TotalDurationNs'])/1e3/nsteps, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
print("total GPU kernel time per step: %.3f ms  (divisor: %d steps = bench.py's \"steps_executed\")" % (tot/1e6/nsteps, nsteps))
# the divisor must be the number of steps the profiled process ran: the vocabulary sweep and its combine run exactly once per step
for r in rows:
    if any(k in r['Name'] for k in ('gen2p_kernel', 'gen2w_kernel', 'gen2_kernel<', 'gen2_combine_kernel')):
        cps = int(r['Calls']) / nsteps
        if abs(cps - 1.0) > 0.02:
            print("WARNING: %s shows %.2f calls/step -- the divisor %d is not the number of steps this process executed; every per-step figure above is off by that factor" % (r['Name'][:60], cps, nsteps))
            sys.exit(3)
