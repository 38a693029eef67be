"""What the N-rank control path of `bench.py --workload cb` costs per step, measured where it can be without GPUs (gloo, 127.0.0.1, one process per
rank): round trips on the shared request cursor (TCPStore `add`), and the byte counts of the two exchanges priced at an assumed xGMI link rate.
    python tools/control_path_cost.py [world=8]
DESIGN.md section 6 quotes the output (profiles/r06_control_path_cost.txt)."""
import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsv-tts-lite_amd")]
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REQ_PER_RANK, TOK_PER_REQ, STEP_S, LINK_GBS = 256, 226, 1.30, 50.0      # the cb32 record: 256 requests, 57.9k tokens, ~1.3 s per step per GPU


def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gsv_tts_lite_amd import engine
    store = engine._default_store()
    n = REQ_PER_RANK * world
    src = engine.RequestSource(list(range(n)), store=store, key="cost/cursor", chunk=2, world=world)
    dist.barrier()
    t0 = time.perf_counter(); got = 0
    while src.next() is not None:
        got += 1
    t_pull = time.perf_counter() - t0
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(200):
        store.add("cost/probe", 0)
    rt = (time.perf_counter() - t0) / 200
    # token exchange as engine.exchange does it on gloo: one all-reduce of the table + a padded all-gather
    tok = torch.zeros(REQ_PER_RANK * TOK_PER_REQ, dtype=torch.int64)
    outs = [torch.empty_like(tok) for _ in range(world)]
    dist.barrier(); t0 = time.perf_counter()
    dist.all_reduce(torch.zeros(3, n, dtype=torch.int64))
    dist.all_gather(outs, tok)
    t_ex = time.perf_counter() - t0
    ret[rank] = (got, t_pull, rt, t_ex)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(worker, args=(world, port, ret), nprocs=world, join=True)
    got = [ret[r][0] for r in range(world)]
    print("world %d, %d requests per rank, every rank pulling as fast as it can (the slot loop pulls one chunk per finished slot):" % (world, REQ_PER_RANK))
    print("  requests pulled per rank: %s (sum %d)" % (got, sum(got)))
    print("  cursor: %.1f us per store.add round trip with %d ranks hammering it; idle store %.1f us" %
          (max(ret[r][1] for r in range(world)) / (max(got) / 2 + 1) * 1e6, world, min(ret[r][2] for r in range(world)) * 1e6))
    rt = max(ret[r][2] for r in range(world))
    n_rt = REQ_PER_RANK / 2 + 53
    print("  per step and rank: %d chunk pulls + ~53 fair_share reads = %d round trips x %.0f us = %.1f ms of host time (%.2f %% of a %.2f s step; the host "
          "issues them between windows, beside the GPU's steps)" % (REQ_PER_RANK / 2, n_rt, rt * 1e6, n_rt * rt * 1e3, n_rt * rt / STEP_S * 100, STEP_S))
    tokb = REQ_PER_RANK * TOK_PER_REQ * 8
    print("  token exchange (gloo here: %.1f ms): %.0f KB per rank, all-gathered; ring over xGMI at %.0f GB/s per link: %.0f us" %
          (max(ret[r][3] for r in range(world)) * 1e3, tokb / 1e3, LINK_GBS, (world - 1) * tokb / (LINK_GBS * 1e9) * 1e6))
    smp = REQ_PER_RANK * TOK_PER_REQ * 2 * 640 * 4          # 2 frames per token, 640 samples per frame (32 kHz), fp32
    print("  samples gathered on rank 0: %.0f MB per rank per step; %d peers -> %.2f GB into rank 0: %.1f ms with every peer on its own link at %.0f GB/s, "
          "%.1f ms if they shared one" % (smp / 1e6, world - 1, (world - 1) * smp / 1e9, smp / (LINK_GBS * 1e9) * 1e3, LINK_GBS, (world - 1) * smp / (LINK_GBS * 1e9) * 1e3))
    worst = n_rt * rt + (world - 1) * tokb / (LINK_GBS * 1e9) + (world - 1) * smp / (LINK_GBS * 1e9)
    best = (world - 1) * tokb / (LINK_GBS * 1e9) + smp / (LINK_GBS * 1e9)
    nb = REQ_PER_RANK * world / 10.0
    imb = (-(-nb // world)) / (nb / world) - 1.0
    print("  vocoder batches of 10 dealt b mod world: %.0f batches, the fullest rank has %.1f %% more than the mean (the stage is ~15 %% of a step)" % (nb, imb * 100))
    print("  weak-scaling bound from the control path alone: %.3f (cursor serial + gather over ONE link) .. %.3f (cursor hidden, one link per peer), "
          "x %.3f for the batch dealing" % (STEP_S / (STEP_S + worst), STEP_S / (STEP_S + best), 1.0 / (1.0 + 0.15 * imb)))
