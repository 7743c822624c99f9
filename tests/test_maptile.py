"""Map-tile assembly: oracle vs the reference's PoseGraph.global_map_query_graph fixture
(CPU), HIP kernel vs the same fixture (GPU)."""
import numpy as np
import pytest
import torch

from conftest import T, load_golden
from oracle import dpm_oracle as O


def _case():
    g = load_golden("maptile.npz")
    order = [int(t) for t in g["tokens"][::256]]
    return g, order


def test_oracle_map_tile_vs_reference():
    g, order = _case()
    assert order == [2, 1, 3, 0]  # BFS order; the non-keyframe (4) and the far scan (5) are left out
    tile = O.map_tile([T(g["key_points"][i]) for i in order], [T(g["SE3"][i]) for i in order], T(g["centering"]))
    np.testing.assert_allclose(tile.numpy(), g["tile"], rtol=0, atol=2e-5)


@pytest.mark.gpu
def test_hip_map_tile_vs_reference():
    from deeppointmap_amd.maptile import MapTileStore, assemble_map_tile
    g, order = _case()
    kp = T(g["key_points"]).to("cuda:0")
    tile = assemble_map_tile(kp, T(g["SE3"]), T(g["centering"]), torch.tensor(order, dtype=torch.int32))
    assert tuple(tile.shape) == (131, 1024)
    assert np.array_equal(tile[:128].cpu().numpy(), g["tile"][:128])       # features: pure copy
    np.testing.assert_allclose(tile[128:].cpu().numpy(), g["tile"][128:], rtol=0, atol=2e-5)
    store = MapTileStore("cuda:0", capacity=2)  # forces a grow
    for i in range(6):
        store.put(100 + i, T(g["key_points"][i]))
    t2, tok = store.tile([100 + i for i in order], [T(g["SE3"][i]) for i in order], T(g["centering"]))
    assert torch.equal(t2, tile) and tok.tolist() == [100 + i for i in order for _ in range(256)]
