import re, subprocess, time
# what the kernel takes to tear a process down, by what it holds: (GB device untouched, GB device written, GB pinned)
for args in (["0", "0", "0"], ["56", "0", "0"], ["0", "56", "0"], ["0", "0", "0.6"], ["28", "28", "0.6"], ["0", "0", "0.1"], ["0", "0", "0"]):
    res = []
    for rep in range(3):
        time.sleep(2.0)
        p = subprocess.run(["bin/exit_cost_probe"] + args, stdout=subprocess.PIPE)
        t1 = time.time()
        m = re.search(r"exit_at ([0-9.]+)", p.stdout.decode())
        res.append(t1 - float(m.group(1)))
    print("device untouched %s GB, written %s GB, pinned %s GB: _exit -> process gone %s s" % (args[0], args[1], args[2], " ".join("%.3f" % x for x in res)), flush=True)
