"""Step time of configs[1] with a measurement build of libbdx.so (variants/*.so): python tools/variant_bench.py --lib variants/x.so"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", required=True)
    ap.add_argument("--length", type=int, default=50_000_000)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--mode", type=int, default=0)
    a = ap.parse_args()
    import breakdancer_amd._lib as lib
    lib.LIB_PATH = os.path.abspath(a.lib)
    import torch
    import breakdancer_amd.api as bda
    from breakdancer_amd.api import LibraryConfig, Options
    from breakdancer_amd.synth import LIB_C2, make_chromosome
    d = make_chromosome(length=a.length, seed=1)
    n = len(d["pos"])
    dev = torch.device("cuda", 0)
    tens = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
    bd = bda.BreakDancer(Options(), [LibraryConfig(**LIB_C2)], 1, ntids=1, max_read_window_size=200, device=0)
    bd.set_device_reads({k: t.data_ptr() for k, t in tens.items()}, n)
    bd.set_enqueue_ahead(a.mode)
    for _ in range(10):
        bd.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        bd.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    svs = bd.svs()
    print("%s mode %d: %.4f ms/step, %d svs, walk split %s" % (os.path.basename(a.lib), a.mode, dt / a.steps * 1e3, len(svs["score"]) if isinstance(svs, dict) else len(svs), bd.walk_split()))
    bd.close()


if __name__ == "__main__":
    main()
