"""Stand-in for bench.py in tests/test_launch.py: the same start-up (plain invocation -> dynamorph_amd.launch -> N ranks
over gloo on the CPU), a real all-reduce, ONE JSON line from rank 0, library-style chatter on stdout beside it."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--fail-rank", type=int, default=-1)
    args = ap.parse_args()
    from dynamorph_amd import launch
    if args.gpus > 1 and not launch.launched():
        launch.check_devices(args.gpus)
        sys.exit(launch.self_launch(__file__, sys.argv[1:], args.gpus))
    import torch
    from dynamorph_amd import dist as D
    rank, world, local = D.init_from_env(backend="gloo")
    if rank == args.fail_rank:
        sys.exit(7)
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        torch.distributed.all_reduce(t)
    slowest = D.max_over_ranks(float(rank))
    print(f"rank {rank}: chatter on stdout that is not the result line", flush=True)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "sum": t.item(), "max_rank": slowest, "self_launched": os.environ.get("DM_SELF_LAUNCHED")}),
              flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
