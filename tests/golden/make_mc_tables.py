"""Fixture generator: the reference's marching-cubes lookup tables as one .npz.

    python tests/golden/make_mc_tables.py     (needs /root/reference; the GPU box does not have it)

surface_render_data/polygon_counts.txt (256 numbers: triangles per corner configuration) and
polygon_edge_indices.txt (256 x 15 numbers: their edge indices, 255 = none) are DATA files the reference
loads at run time (marching_cubes.h:30-33); they are the classic Lorensen-Cline / Bourke tables.  The tests
and tools read this copy; a deployment loads the originals the same way the reference does
(fluid_flow_sections_amd.hpp: MarchingCubesBuffers::loadData)."""
import os
import numpy as np

REF = "/root/reference/surface_render_data"
HERE = os.path.dirname(os.path.abspath(__file__))
counts = np.array(open(os.path.join(REF, "polygon_counts.txt")).read().split(), dtype=np.uint32)
edges = np.array(open(os.path.join(REF, "polygon_edge_indices.txt")).read().split(), dtype=np.uint32)
assert counts.shape == (256,) and edges.shape == (256 * 15,)
# consistency of the two tables: 3 indices per triangle, 255 beyond them
for cfg in range(256):
    row = edges[15 * cfg:15 * cfg + 15]
    assert np.all(row[:3 * counts[cfg]] < 12) and np.all(row[3 * counts[cfg]:] == 255), cfg
np.savez_compressed(os.path.join(HERE, "marching_cubes_tables.npz"), counts=counts, edge_indices=edges)
print("wrote marching_cubes_tables.npz:", int(counts.sum()), "triangles over 256 configurations")
