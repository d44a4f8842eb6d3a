import os, subprocess, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from fuzzgen import make_case
from breakdancer_amd.bamwrite import write_bam_records
import tempfile
tmp = tempfile.mkdtemp()
seed = 0
rng = np.random.default_rng(40 + seed)
cfg, streams, targets = make_case(860 + seed, n_pairs=int(rng.integers(1500, 6000)))
cfg1 = "".join(l + "\n" for l in cfg.splitlines() if "map:a.bam" in l)
st = streams[0]
recs = [dict(tid=st["tid"][i], pos=st["pos"][i], mtid=st["mtid"][i], mpos=st["mpos"][i], isize=st["isize"][i], flag=st["flag"][i],
             qlen=st["qlen"][i], mapq=int(st["bdqual"][i]), rg=st["rg"][i], name="read%d" % int(st["name_id"][i])) for i in range(len(st["tid"]))]
write_bam_records(os.path.join(tmp, "a.bam"), recs, targets, rgs=("rg1", "rg2", "rg3"), seed=seed, index=True)
open(os.path.join(tmp, "cfg"), "w").write(cfg1)
print("targets", targets, "n", len(recs), np.bincount(np.asarray(st["tid"]) + 1))
for label, env in (("device", dict(BDX_GPUS="0,0", BDX_TIMING="1")), ("host", dict(BDX_GPUS="0,0", BDX_TIMING="1", BDX_DECODE="host"))):
    p = subprocess.run([os.path.abspath("bin/breakdancer-max"), "-y", "-1", os.path.join(tmp, "cfg")], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
    print(label, p.returncode)
    print("\n".join(p.stdout.decode().splitlines()[:8]))
    print("\n".join(l for l in p.stderr.decode().splitlines() if "sharded" in l or "decode:" in l or "rror" in l)[:1500])
