"""Worker for the multi-process tests.  Launched by torch.distributed.run with N processes.

mode "comm" (CPU, gloo): exercises breakdancer_amd.shard's restatement of the native sharded run's rules -- planning, counter
all-reduce, per-chromosome bases, the all-to-all of the inter-chromosomal join entries to the rank of the LATER chromosome,
the join there (own + foreign entries), the merge of the ranks' tables by order key on rank 0, the taint rule -- with a numpy
stand-in for the join so that the result can be checked against a single-process computation.  No GPU, no libbdx compute calls.

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
    """reference join on a set of entries: pairs = two entries with one key; second-observed = larger order (stream order)"""
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
    # join entries: pairs whose mates live on different chromosomes (hence possibly different ranks).  Stream order follows the
    # chromosome: a read on a later chromosome is observed later.
    owner = np.full(ntid, -1, np.int64)
    for r, tids in enumerate(plan):
        owner[tids] = r
    npairs = 4000
    keys = rng.integers(1, 2**63, npairs, dtype=np.int64).astype(np.uint64)
    ta = rng.integers(0, ntid, npairs)
    tb = (ta + rng.integers(1, ntid, npairs)) % ntid          # the mate's chromosome: another one
    ent_all = np.zeros(2 * npairs, shard.ENTRY_DTYPE)
    ent_all["key"] = np.concatenate([keys, keys])
    ent_all["tid"] = np.concatenate([ta, tb])
    ent_all["mtid"] = np.concatenate([tb, ta])
    ent_all["order"] = ent_all["tid"].astype(np.uint32) * np.uint32(100000) + rng.permutation(2 * npairs).astype(np.uint32)
    ent_all["region"] = ent_all["tid"] * 100 + rng.integers(0, 100, 2 * npairs)
    ent_all["meta"] = 8 | (rng.integers(0, 3, 2 * npairs) << 8)
    ent_all["isize"] = rng.integers(0, 5000, 2 * npairs)
    mine_ent = ent_all[np.isin(ent_all["tid"], mine)]
    chunks, stay = shard.route_entries(mine_ent, owner, rank, world)
    assert len(chunks[rank]) == 0                                      # nothing travels to oneself
    recv = comm.alltoall_bytes(chunks)
    foreign = np.concatenate([np.frombuffer(r.tobytes(), shard.ENTRY_DTYPE) for r in recv])
    # a foreign entry's mate chromosome is one of mine and comes later than its own: it is the first-observed mate of a pair
    assert np.isin(foreign["mtid"], mine).all() and (foreign["mtid"] > foreign["tid"]).all()
    sent = sum(len(c) for c in chunks) // shard.ENTRY_DTYPE.itemsize
    assert int(comm.allreduce_sum(np.array([sent], np.uint64))[0]) <= npairs   # at most one mate of a pair travels
    # every pair whose SECOND mate is mine can be joined here: own entries + foreign ones
    here = np.concatenate([mine_ent, foreign])
    part = {k: v for k, v in numpy_join(here).items() if k[1] // 100 in mine}   # keyed by the later region = the second mate's
    gathered = comm.gather_obj(part, root=0)
    # the tables of the ranks, each sorted by order key, merge into the genome's order (start vertices belong to one rank each)
    nreg = ntid * 100
    starts_all = np.sort(rng.choice(nreg, 600, replace=False))
    T_all = np.where(rng.random(600) < 0.2, (starts_all // 101 + 1) * 101, starts_all)   # some rows are placed at a later window's first vertex
    keys_all = shard.order_key(T_all, starts_all)
    mine_rows = np.isin(starts_all // 100, mine)
    my_keys = np.sort(keys_all[mine_rows])
    tables = comm.gather_obj(my_keys, root=0)
    if rank == 0:
        merged = {}
        for d in gathered:
            for k, v in d.items():
                c = merged.setdefault(k, [0, 0])
                c[0] += v[0]
                c[1] += v[1]
        truth = numpy_join(ent_all)
        assert merged == truth and sum(v[0] for v in truth.values()) == npairs
        order = shard.merge_by_key(tables)
        got = np.array([tables[q][i] for q, i in order], dtype=np.uint64)
        assert (got == np.sort(keys_all)).all()
        # taint: a group between regions of two ranks taints both
        own_reg = owner[np.arange(nreg) // 100]
        t = shard.taint_regions([(k[0], k[1]) for k in truth], own_reg, nreg)
        for lo, hi, _, _ in truth:
            assert (t[lo] == 1 and t[hi] == 1) == (own_reg[lo] != own_reg[hi]) or (t[lo] and t[hi])
        json.dump({"ok": True, "world": world, "groups": len(truth)}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    {"comm": mode_comm}[sys.argv[1]](sys.argv[2])
