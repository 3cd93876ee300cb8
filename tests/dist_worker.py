"""One rank of tests/test_gpu_dist.py: builds the same engine on every rank, attaches the data-parallel hooks
(nvfpcc_amd/dist.py) and runs NVFPCC.py train's epoch loop (engine.EpochDriver + latent step + all-gather) for a
few epochs, then rank 0 saves the replicated state.  Started as `python tests/dist_worker.py OUT N B EPOCHS` with the
torchrun environment (RANK / WORLD_SIZE / MASTER_*); NVF_DIST_BACKEND=gloo + NVF_DEVICE_OVERRIDE=0 let several
ranks share the one GPU of a test box (RCCL refuses two ranks on one device)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, N, B, epochs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    from nvfpcc_amd import dist as nd, network
    from nvfpcc_amd.engine import TrainEngine, EpochDriver
    from nvfpcc_amd.model import Net
    from nvfpcc_amd.seeds import synthetic_seed
    from nvfpcc_amd.synth import make_blocks
    from tests.golden_inputs import CONFIGS, perturb_state_, make_emb
    rank, local_rank, world = nd.init()
    dev = torch.device("cuda", int(os.environ.get("NVF_DEVICE_OVERRIDE", local_rank)))
    torch.cuda.set_device(dev)
    cfg = CONFIGS["S"]
    network.reset_seed(synthetic_seed())
    network.set_noise_seed(0, 0)
    net = Net(None, "Gaussian", cfg["ch"], ",".join(str(c) for c in cfg["channels"]), verbose=False)
    sd = net.state_dict()
    perturb_state_(sd, cfg["param_seed"])
    net.load_state_dict(sd)
    net = net.to(dev)
    gts, dists = make_blocks(N)
    eng = TrainEngine(net, torch.from_numpy(gts).float().to(dev), torch.from_numpy(dists).float().to(dev),
                      n_points_total=917 * 936.0, emb=make_emb(N, cfg["ch"], cfg["emb_seed"]), seed=0, lmbda=200.0,
                      w1=10.0, w2=57.0, lr=1e-3, wemb=5.0)
    nd.attach(eng, world)
    drv = EpochDriver(eng, B, rank, world, use_graph=os.environ.get("NVF_TRAIN_GRAPH", "1") != "0")
    lo, hi = nd.shard_range(N, rank, world)
    rng = np.random.default_rng(17)            # the epoch order every rank derives from the shared seed
    stats, grads = [], []
    for epoch in range(epochs):
        q = 1 if epoch == 0 else 2
        drv.run(rng.permutation(N), q)
        grads.append(eng.flat_g.clone().cpu())              # the last mini-batch's all-reduced gradient
        if hi > lo:
            eng.latent_step(q, lo, hi)
        else:
            eng.noise_step += 1
        nd.allgather_rows_(eng.emb, rank, world)
        stats.append(eng.read_epoch_stats(reduce=nd.allreduce_sum_ if world > 1 else None, world=world))
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"flat_p": eng.flat_p.cpu(), "emb": eng.emb.cpu(), "stats": np.stack(stats), "grads": grads,
                    "noise_step": eng.noise_step, "opt_step": eng.opt_step, "world": world,
                    "graphs": sorted(drv.graphs)}, out)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
