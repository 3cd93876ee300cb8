"""Drop-in module path of the reference (`from gdn_3d import GDN3d, IGDN3d`): gfx950 implementation."""
from nvfpcc_amd.gdn_3d import GDN3d, IGDN3d  # noqa: F401
