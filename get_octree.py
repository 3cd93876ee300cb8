#!/usr/bin/env python3
"""`./get_octree input.ply origins.txt subtree.txt` of the reference (get_octree.cpp:751-800; README step 1a),
as a script: level-5 cube origins in the reference's traversal order and the shallow occupancy bit string."""
import sys

from nvfpcc_amd.preprocess import octree_level5, read_ply_xyz, write_origins_txt

if __name__ == "__main__":
    origins, subtree = octree_level5(read_ply_xyz(sys.argv[1]))
    write_origins_txt(sys.argv[2], origins)
    with open(sys.argv[3], "w") as f:
        f.write(subtree)
