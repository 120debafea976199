"""One rank of the two-process peer-transport test (launched by torch.distributed.run; both ranks on GPU 0).
Not collected by pytest."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import pyref
    from oracle import Oracle
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    D = pkg.distributed
    for p in (pyref.GOLDILOCKS, 389):
        o = Oracle(p)
        ctx = pkg.Context(pkg.Field(p), device=0)
        ctx.set_option("peer_spin_ms", 60000)   # a failure detector: generous (a cold box stalls ranks for many seconds)
        D.attach_peer(ctx, rank, world)
        assert ctx.rank_world() == (rank, world)
        # grid_sharded 1: the shards go on with five-round passes (cells exchanged inside the kernel) down to one entry;
        # 0: two-round passes with the exchange, gather at tail_log, unsharded tail
        for n, tail_log, gs in [(1, 0, 1), (2, 0, 1), (5, 0, 1), (12, 0, 1), (12, 5, 0), (12, 0, 0), (16, 12, 1), (16, 12, 0), (20, 16, 1),
                                (20, 16, 0), (22, 16, 1)]:
            if n < world.bit_length() - 1:
                continue            # fewer entries than ranks
            ctx.set_option("tail_log", tail_log)
            ctx.set_option("grid_sharded", gs)
            start, length = D.shard_range(n, rank, world)
            nl = length.bit_length() - 1
            a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, nl, start=start)
            b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, nl, start=start)
            g = pkg.matrix_multiplication.G(a, b)
            assert g.num_vars() == n
            ctx.set_option("time_kernels", 1)
            ctx.launch_log(reset=True)
            c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
            log = ctx.launch_log(reset=True)
            ctx.set_option("time_kernels", 0)
            if gs and nl >= 6:      # the shard's own variables are served five at a time, then one pass on the gathered table
                assert [r["kind"] for r in log].count("grid_pass") >= 2 and log[-1]["log_in"] <= 5 + world.bit_length() - 1, log
            oa, ob = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
            ref = o.prove(oa, ob, ch)
            assert ref["status"] == 0
            assert c1 == ref["c_1"], (p, n, tail_log, "c_1")
            assert np.array_equal(evals, ref["evals"]), (p, n, tail_log, "round polynomials")
            assert g.evaluate([int(x) for x in ch]) == ref["final_eval"], (p, n, "evaluate")
            assert g.hypercube_sum() == ref["c_1"]
            if nl >= 1:
                assert g.round_evals() == [int(x) for x in ref["evals"][0]]
            del a, b, g
        # ranks fed different challenges must fail loudly (digest in the exchange), on every rank
        n = 10
        start, length = D.shard_range(n, rank, world)
        nl = length.bit_length() - 1
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, nl, start=start)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, nl, start=start)
        g = pkg.matrix_multiplication.G(a, b)
        ctx.set_option("tail_log", 0)
        ctx.set_option("grid_sharded", 1)
        try:
            pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R + (7 if rank == 1 else 0))
            raise AssertionError("different challenges were accepted")
        except pkg.SumcheckHipError as e:
            assert e.code == 5 and "different challenges" in str(e), e
        dist.barrier()
        del a, b, g
        ctx.close()
    print("PEER-OK rank %d" % rank, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
