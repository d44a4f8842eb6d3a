"""Worker of tests/test_gpu_rccl.py: one rank of a REAL multi-GPU sharded run (one process per GPU, RCCL communicator).
Launched by torch.distributed.run; every rank rebuilds the same seeded fuzz case, loads the chromosomes bdx_dist_plan gives it,
all call bdx_dist_run; rank 0 compares the whole-genome result with the oracle and writes the verdict."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(out_path):
    import torch
    import torch.distributed as dist
    from breakdancer_amd import dist as D
    from breakdancer_amd.api import LibraryConfig
    from fuzzgen import clash_names, make_case
    from helpers import make_opts
    from runner import compare, oracle_case, product_options, split_by_tid
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl")
    verdicts = []
    for seed, kw, clash in ((131, dict(transchr_rearrange=1, min_read_pair=1), False), (132, dict(), False), (133, dict(cn_lib=1, print_af=1), False),
                            (946, dict(min_read_pair=1, buffer_size=1), True)):
        cfg, streams, targets = make_case(seed)
        if clash:
            streams = clash_names(streams, seed, frac=0.04)
        run = oracle_case(cfg, streams, targets, make_opts(score_threshold=-1, **kw))
        libs = [LibraryConfig(*[float(x) for x in run.lib_f[i]], min_mapping_quality=int(run.lib_i[i, 0]),
                              bam_file_index=int(run.lib_i[i, 1]), name=run.lib_names[i]) for i in range(run.nlibs)]
        d = D.DistRun.from_process_group(product_options(run.opts), libs, run.nbams, len(targets), run.w0, local)
        chroms = split_by_tid(run.merged_soa())
        rank_of = D.plan([len(chroms[t]["tid"]) if t in chroms else 0 for t in range(len(targets))], world)
        for tid, arrs in chroms.items():
            if rank_of[tid] == rank:
                d.chromosome(tid).push_reads(arrs)
        d.run()
        ex = d.exchange()
        coll = d.collectives()
        if rank == 0:
            res = d.result()
            compare(run, res, check_cls=False)
            verdicts.append(dict(seed=seed, svs=int(run.n_svs), replayed=bool(res.was_replayed()), ctx_sent_rank0=ex["ctx_records_sent"], collectives=coll))
        dist.barrier()
        d.close()
    if rank == 0:
        json.dump(dict(ok=True, world=world, cases=verdicts), open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
