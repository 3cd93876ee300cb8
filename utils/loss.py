"""Drop-in module path of the reference (`from utils.loss import ...`): gfx950 implementation."""
from nvfpcc_amd.loss import (get_focal_dense, get_acc_dense, get_surf_dual_dense, get_surf_focal_dense, get_sse1,  # noqa: F401
                             get_se, get_surface_loss_dense)
