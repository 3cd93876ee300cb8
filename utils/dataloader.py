"""Drop-in module path of the reference (`from utils.dataloader import LoadedVoxelDataset`)."""
from nvfpcc_amd.dataloader import LoadedVoxelDataset  # noqa: F401
