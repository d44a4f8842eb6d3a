#!/usr/bin/env python3
"""How far does the host side of the BAM feed scale when N GPUs' feeders share one host?  

N concurrent bin/bdx-feed-probe instances (N = 1, 2, 4, 8) on one file in the page cache, the box's usable CPUs divided between them
(and, second table, every instance with all of them: oversubscribed).  The leg that matters is page cache -> pinned staging (reader
threads' memcpy out of the file's mapping): host memory bandwidth and cores, shared by all feeders of a node.  The pinned -> HBM legs
of the N instances all go through THIS box's one GPU link, so their aggregate is that link's ceiling, not a node's -- a node has one link
per GPU -- and is reported only to say so.  usage: feed_scaling.py <file> [GiB per instance = 4]  -> one JSON line per N, then a summary."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def usable_cpus():
    """hardware threads this process may run on, capped by the cgroup CPU quota (a container may see 256 threads and own 16)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def run(path, gib, n_inst, threads, how=1):
    ps = [subprocess.Popen([os.path.join(ROOT, "bin", "bdx-feed-probe"), path, str(gib), str(threads), "12", str(how)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
          for _ in range(n_inst)]
    outs = []
    for p in ps:
        o, e = p.communicate(timeout=600)
        if p.returncode != 0:
            return {"error": e.decode()[-300:]}
        outs.append(json.loads(o.decode().strip().splitlines()[-1]))
    host = [o["page_cache_to_pinned_gb_s"] for o in outs]
    both = [o["both_pipelined_gb_s"] for o in outs]
    return {"instances": n_inst, "threads_per_instance": threads, "page_cache_to_pinned_gb_s": {"per_instance_min": min(host), "per_instance_mean": sum(host) / len(host), "aggregate": sum(host)},
            "both_pipelined_through_one_gpu_link_gb_s": {"per_instance_mean": sum(both) / len(both), "aggregate": sum(both)}}


def main():
    path = sys.argv[1]
    gib = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    cpus = usable_cpus()
    need = 44.4   # GB/s of file one GPU's inflate kernel takes (random-base level-1 file; a realistic level-6 file needs ~30)
    rows = []
    for oversub in (False, True):
        for n in (1, 2, 4, 8):
            t = cpus if oversub else max(1, cpus // n)
            r = run(path, gib, n, t)
            r["usable_cpus"] = cpus
            r["oversubscribed"] = oversub
            rows.append(r)
            print(json.dumps(r), flush=True)
    ok = [r for r in rows if "error" not in r and not r["oversubscribed"]]
    below = [r["instances"] for r in ok if r["page_cache_to_pinned_gb_s"]["per_instance_mean"] < need]
    print(json.dumps({"summary": "with %d usable CPUs the host leg of the feed falls below %.1f GB/s per GPU (one inflate kernel's appetite on the level-1 file) from N = %s on"
                                 % (cpus, need, below[0] if below else "> 8"), "need_gb_s_per_gpu": need}))


if __name__ == "__main__":
    main()
