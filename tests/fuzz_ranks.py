#!/usr/bin/env python3
"""Rank-level fuzz campaign (not collected by pytest; tests/test_gpu_multirank.py::test_rank_fuzz_seeds runs 12 seeds per world
size): `tests/dist_worker.py fuzz:<seed>:<count>` over 2 ... 8 rank processes sharing one GPU through tests/mock_rccl.

    python tests/fuzz_ranks.py [seeds per launch] [launches per world size] [first seed]"""
import os
import pathlib
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_gpu_multirank as T


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 400000
    total, failed = 0, []
    for world in (2, 3, 4, 5, 8):
        for _ in range(launches):
            with tempfile.TemporaryDirectory() as d:
                try:
                    results = T._run_ranks_once(world, f"fuzz:{seed}:{count}", pathlib.Path(d), mock=True)     # (a campaign repeats nothing)
                    done = results[0]["seeds"]
                    total += len(done)
                    print(f"world {world}: seeds {seed}..{seed + count - 1}: {len(done)} systems ok", flush=True)
                except AssertionError as e:
                    failed.append((world, seed))
                    print(f"world {world}: seeds {seed}..{seed + count - 1}: FAILED\n{str(e)[-2500:]}", flush=True)
            seed += count
    print(f"{total} systems over ranks, failing launches: {failed}")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
