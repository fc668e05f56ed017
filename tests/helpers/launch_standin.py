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
    ap.add_argument("--hang-rank", type=int, default=-1, help="this rank never finishes (and ignores SIGTERM)")
    ap.add_argument("--launch-timeout", type=float, default=None)
    ap.add_argument("--pid-dir", default=None, help="every rank writes <pid-dir>/rank<r>.pid")
    ap.add_argument("--ragged-batch", type=int, default=13)
    args = ap.parse_args()
    from dynamorph_amd import launch
    if args.gpus > 1 and not launch.launched():
        launch.check_devices(args.gpus)
        launch.GRACE_SECONDS = 3.0
        sys.exit(launch.self_launch(__file__, sys.argv[1:], args.gpus, timeout=args.launch_timeout))
    import time
    import torch
    from dynamorph_amd import dist as D
    rank, world, local = D.init_from_env(backend="gloo")
    if args.pid_dir:
        with open(os.path.join(args.pid_dir, f"rank{rank}.pid"), "w") as f:
            f.write(str(os.getpid()))
    if rank == args.fail_rank:
        sys.exit(7)
    if rank == args.hang_rank:
        import signal
        signal.signal(signal.SIGTERM, signal.SIG_IGN)      # a rank stuck in a collective does not answer SIGTERM either
        while True:
            time.sleep(1.0)
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        torch.distributed.all_reduce(t)
    slowest = D.max_over_ranks(float(rank))
    # what bench.py's line carries for N > 1, from the same helpers: every rank's own time, the collective's world and
    # backend, and the gradient weights of a ragged global batch (train()'s plan)
    rank_ms = D.gather_rank_values(1.0 + rank)
    weights = D.gather_rank_values(D.shard_weight(args.ragged_batch, rank, world))
    print(f"rank {rank}: chatter on stdout that is not the result line", flush=True)
    if rank == 0:
        rec = {"n_gpus": world, "sum": t.item(), "max_rank": slowest, "self_launched": os.environ.get("DM_SELF_LAUNCHED")}
        if world > 2:
            rec.update(collective={"world": world, "backend": torch.distributed.get_backend(), "rank_ms_per_step": rank_ms},
                       shard_weights=weights)
        print(json.dumps(rec), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
