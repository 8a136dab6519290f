"""One rank of tests/test_two_ranks_one_gpu.py: launched twice by torch.distributed.run, BOTH ranks on cuda:0 (a gloo
group -- RCCL refuses two ranks on one device -- and the host-staged reward-log gather).  Rank r steps replicas
[r E, (r + 1) E) of a 2 E batch with the real HIP backend: env_offset = r E keys the generators and the synthetic policy
by GLOBAL replica id.  Writes rank<r>.npz into the directory given on the command line: the final state of its shard and,
on the learner rank, every gathered (reward, done) block."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

STATE_KEYS = ("stone", "wood", "house_owner", "loc_r", "loc_c", "inv_res", "inv_coin", "esc_coin", "labor", "util", "mt", "mt_pos",
              "timestep", "completions", "cda_bids", "cda_n_bids", "tax_cycle_pos", "rewards_a", "rewards_p", "done",
              "obs_a_flat", "obs_a_action_mask", "obs_p_flat")


def main():
    out_dir, E, steps, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import torch
    import torch.distributed as dist

    from ai_economist_amd.sharding import RewardLogGather, dist_info
    from helpers import C2, make_env

    rank, _local, world = dist_info()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    env = make_env(dict(C2, episode_length=9), n_envs=E, device="cuda:0", env_offset=rank * E)
    env.seed(11)
    env.reset()
    be = env.backend
    be.set_auto_reset(True)
    g = RewardLogGather(be, steps_per_gather=K, keep=True)
    assert g.host_staged and g.collective
    cur = be.sample_random_actions(77, rank * E, slot=0)
    slot = 0
    for _ in range(steps):
        cur = be.step_sample_next(cur[0], cur[1], 77, rank * E, next_slot=slot ^ 1)
        slot ^= 1
        g.after_step()
    g.finish()
    torch.cuda.synchronize()
    out = {k: be.tensors[k].cpu().numpy() for k in STATE_KEYS if k in be.tensors}
    if rank == 0:
        for i, blk in enumerate(g.received):
            out["block%d" % i] = blk.cpu().numpy()
        out["n_blocks"] = np.int64(len(g.received))
        out["n_collectives"] = np.int64(g.n_collectives)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
