import re, subprocess, time
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
for args in (["0", "0.5"], ["1", "0.5"], ["2", "0.5"], ["0", "0.5"], ["1", "0.5"]):
    for rep in range(2):
        time.sleep(1.5)
        p = subprocess.run(["bin/pin_probe"] + args, stdout=subprocess.PIPE)
        t1 = time.time()
        out = p.stdout.decode().strip()
        m = re.search(r"exit_at ([0-9.]+)", out)
        print(out.split(" exit_at")[0], "| _exit -> gone %.3f s" % (t1 - float(m.group(1))) if m else out, flush=True)
