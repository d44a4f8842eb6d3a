"""Worker for the multi-process tests.  Launched by torch.distributed.run with N processes.

mode "comm" (CPU, gloo): exercises breakdancer_amd.shard's collectives -- planning, counter all-reduce, base
all-gather, the all-to-all routing of join entries, the gather to rank 0 -- with a numpy stand-in for the join so that
the result can be checked against a single-process computation.  No GPU, no libbdx compute calls.

(The native multi-rank run needs GPUs: tests/test_gpu_sharded.py drives it with the ranks as threads on one device.)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def numpy_join(ent):
    """reference join on the routed entries: pairs = two entries with one key; second-observed = larger order"""
    out = {}
    if len(ent) == 0:
        return out
    o = np.argsort(ent["key"], kind="stable")
    e = ent[o]
    same = np.flatnonzero(e["key"][1:] == e["key"][:-1])
    for i in same:
        a, b = e[i], e[i + 1]
        if a["order"] > b["order"]:
            a, b = b, a
        k = (int(a["region"]), int(b["region"]), int(b["meta"]) & 15, (int(b["meta"]) >> 8) & 255)
        c = out.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += int(b["isize"])
    return out


def mode_comm(out_path):
    import torch.distributed as dist
    from breakdancer_amd import shard
    dist.init_process_group("gloo")
    comm = shard.TorchComm()
    rank, world = comm.rank, comm.world
    rng = np.random.default_rng(7)  # same stream on every rank: everybody can compute the global truth
    ntid = 7
    counts = {t: int(rng.integers(1000, 90000)) for t in range(ntid)}
    plan = shard.plan_chromosomes(counts, world)
    assert sorted(t for b in plan for t in b) == list(range(ntid))
    loads = [sum(counts[t] for t in b) for b in plan]
    assert max(loads) - min(loads) <= max(counts.values())
    # per-tid synthetic "pass 1" payloads
    ncnt, nbams, nkeys = 26, 2, 2
    cnt = {t: rng.integers(0, 1000, ncnt).astype(np.uint64) for t in range(ntid)}
    ref = {t: rng.integers(10**6, 3 * 10**8, nbams).astype(np.uint64) for t in range(ntid)}
    tot = {t: rng.integers(0, 5000, 2 + nkeys).astype(np.int64) for t in range(ntid)}
    mine = plan[rank]
    red = comm.allreduce_sum(np.concatenate([sum((cnt[t] for t in mine), np.zeros(ncnt, np.uint64)),
                                             sum((ref[t] for t in mine), np.zeros(nbams, np.uint64))]))
    assert (red[:ncnt] == sum(cnt.values())).all() and (red[ncnt:] == sum(ref.values())).all()
    assert shard.covered_from(red[ncnt:]) == int(max(int(x) for x in red[ncnt:])) & 0xFFFFFFFF
    all_tot = {}
    for d in comm.allgather_obj({t: tot[t] for t in mine}):
        all_tot.update(d)
    bases = shard.prefix_bases(all_tot)
    acc = np.zeros(2 + nkeys, np.int64)
    for t in range(ntid):
        assert (bases[t] == acc).all()
        acc += tot[t]
    # join entries: pairs whose mates live on different chromosomes (hence possibly different ranks)
    npairs = 4000
    keys = rng.integers(1, 2**63, npairs, dtype=np.int64).astype(np.uint64)
    ta, tb = rng.integers(0, ntid, npairs), rng.integers(0, ntid, npairs)
    ent_all = np.zeros(2 * npairs, shard.ENTRY_DTYPE)
    ent_all["key"] = np.concatenate([keys, keys])
    ent_all["order"] = rng.permutation(2 * npairs).astype(np.uint32)
    tid_of = np.concatenate([ta, tb])
    ent_all["region"] = tid_of * 100 + rng.integers(0, 100, 2 * npairs)
    ent_all["meta"] = rng.integers(1, 9, 2 * npairs) | (rng.integers(0, 3, 2 * npairs) << 8)
    ent_all["isize"] = rng.integers(0, 5000, 2 * npairs)
    mine_ent = ent_all[np.isin(tid_of, mine)]
    recv = comm.alltoall_bytes(shard.route_entries(mine_ent, world))
    got = np.concatenate([np.frombuffer(r.tobytes(), shard.ENTRY_DTYPE) for r in recv])
    assert (shard.owner_of(got["key"], world) == rank).all()          # routed to the owner
    uk, cnts = np.unique(got["key"], return_counts=True)
    assert (cnts == 2).all()                                           # both mates met on this rank
    part = numpy_join(got)
    gathered = comm.gather_obj(part, root=0)
    if rank == 0:
        merged = {}
        for d in gathered:
            for k, v in d.items():
                c = merged.setdefault(k, [0, 0])
                c[0] += v[0]
                c[1] += v[1]
        truth = numpy_join(ent_all)
        assert merged == truth and sum(v[0] for v in truth.values()) == npairs
        json.dump({"ok": True, "world": world, "groups": len(truth)}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    {"comm": mode_comm}[sys.argv[1]](sys.argv[2])
