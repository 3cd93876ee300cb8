"""LoadedVoxelDataset with the reference's file triple and item semantics
(/root/reference/utils/dataloader.py:152-181), plus a device-resident view used by the trainer:
the grids of every leaf block stay in HBM (4096 blocks x 256 KiB = 1 GiB of 288 GB) instead of
being re-copied from a DataLoader worker each step."""
import numpy as np
import torch


class LoadedVoxelDataset(torch.utils.data.Dataset):
    MAGIC = 2113   # index permutation multiplier (dataloader.py:165-167)

    def __init__(self, origin_fn, gt_fn, dist_fn, shuffle=True):
        super().__init__()
        self.shuffle = shuffle
        self.origins, self.gt_grid, self.dist = (np.load(fn) for fn in (origin_fn, gt_fn, dist_fn))
        self.N_leaf = self.origins.shape[0]      # leaf blocks
        self.N = self.gt_grid.sum()              # occupied voxels = input points (b_net's denominator, NVFPCC.py:162)
        print(f"[data] {self.N_leaf} leaf blocks, {self.N} points")

    def permute(self, idx):
        return (idx * self.MAGIC) % self.N_leaf if self.shuffle else idx

    def __getitem__(self, idx):
        idx = self.permute(idx)
        tidx = torch.tensor(np.array([idx], dtype=np.float32)).long()
        return tidx, torch.from_numpy(self.gt_grid[idx]).float(), torch.from_numpy(self.dist[idx]).float()

    def get_all(self):
        return torch.from_numpy(self.gt_grid).float(), torch.from_numpy(self.dist).float()

    def __len__(self):
        return self.N_leaf

    # ---- MI355X-native feeding: everything resident on the device ----
    def to_device(self, device):
        gt, dist = self.get_all()
        return gt.to(device), dist.to(device)

    def epoch_order(self, epoch, shuffle_loader, seed=0):
        """Block ids in the order one epoch visits them: DataLoader order (sequential, or a seeded
        permutation when --shuffle is truthy) composed with the dataset's own index permutation."""
        if shuffle_loader:
            g = torch.Generator().manual_seed(seed * 1000003 + epoch)
            order = torch.randperm(self.N_leaf, generator=g).numpy()
        else:
            order = np.arange(self.N_leaf)
        return np.array([self.permute(int(i)) for i in order], dtype=np.int64)
