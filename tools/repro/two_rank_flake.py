"""Characterises the intermittent failure of tests/test_gpu_dist_graph.py::test_two_rank_overlapped_split_replay_bf16_one_optimizer_step
(2 of ~10 suite runs in round 5: backbone buckets of the graph run 1.4 - 1.9x the eager run's, cosine 0.92 - 0.98).
For each of P pre-trained points: R graph runs and R eager runs of the test's own 2-rank worker; prints every bucket's norm per
run and the pairwise relative L2 between ALL runs (graph-graph, eager-eager, graph-eager), so one sees which side moves.

    python tools/repro/two_rank_flake.py [P] [R]        (one GPU; ~25 s per pre-train + ~12 s per 2-rank run)
"""
import os
import shutil
import sys
import tempfile

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from tests import test_gpu_dist_graph as t
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    for p in range(P):
        tmp = tempfile.mkdtemp(prefix="flake")
        t._pretrain(tmp)
        runs = []
        for r in range(R):
            for use_graph in (True, False):
                mp.spawn(t._worker_overlap, args=(2, t._free_port(), tmp, use_graph), nprocs=2, join=True)
                tag = "g" if use_graph else "e"
                info = torch.load(os.path.join(tmp, f"o{tag}0.pt"))
                runs.append((f"{tag}{r}", [b.double() for b in info["buckets"]], info["loss"], info["overlap_check"]))
        print(f"== pre-trained point {p}", flush=True)
        for name, bk, loss, chk in runs:
            print(f"  {name}: loss {loss:.5f} overlap_check {chk} norms", [round(float(b.norm()), 4) for b in bk], flush=True)
        for i in range(len(runs)):
            for j in range(i + 1, len(runs)):
                rel = [round(float((a - b).norm() / b.norm()), 3) for a, b in zip(runs[i][1], runs[j][1])]
                print(f"  rel {runs[i][0]} vs {runs[j][0]}:", rel, flush=True)
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
