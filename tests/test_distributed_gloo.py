"""Multi-rank path on CPU: world_size 2 and 3 over gloo.  Each rank produces the raw sums of ITS interleaved
row bands (here taken from the oracle's raw per-pixel sums -- the renderer itself needs a GPU), the ranks
exchange them with path_tracer_ocaml_amd.distributed.BandGather (pre-allocated buffers, one group of sends into
rank 0, banded layout kept as it is), and rank 0's gathered buffer, read through the film kernel's row map, must be
exactly the single-process raw sums.  Also checks the band bookkeeping for 2..8 ranks: the Python layout, the row
map (pt_band_row's twin) and the C ABI's ptx_local_rows / ptx_global_row agree."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, height, width, raw_path, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import path_tracer_ocaml_amd as P
    from path_tracer_ocaml_amd import distributed as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.load(raw_path)
    params = P.render_params(width, height, 1, 1, band_rows=D.BAND_ROWS, band_first=rank, band_step=world)
    rows = P.local_rows(params)
    layout = D.band_layout(height, world)
    assert rows == len(layout[rank])
    assert [P.global_row(params, k) for k in range(rows)] == layout[rank].tolist()
    bg = D.BandGather(height, width, rank, world, torch.device("cpu"))
    assert bg.local_rows == rows and bg.pad_rows >= rows
    for step in range(2):  # the second step re-uses every buffer: nothing may be left over from the first
        ptr = bg.part.data_ptr()
        bg.part[:rows] = torch.from_numpy(full[layout[rank]] * (step + 1))  # what ptx_render_raw_device would have written
        got = bg.gather()
        assert bg.part.data_ptr() == ptr
        if rank == 0:
            assert got.data_ptr() == bg.gathered.data_ptr() and got.shape == (world, bg.pad_rows, width, 3)
        else:
            assert got is None
    if rank == 0:
        np.save(out_path, D.ungather(got.numpy(), height, world) / 2.0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 150), (3, 100), (2, 31)])
def test_band_gather_over_gloo(oracle, tmp_path, world, height):
    import torch.multiprocessing as mp
    width = 64
    d = oracle.desc_shirley(width, height)
    raw = oracle.Scene(d.ptr, d).render(width, height, 2, 4, threads=4, want_raw=True)["raw"]
    raw_path, out_path = str(tmp_path / "raw.npy"), str(tmp_path / "out.npy")
    np.save(raw_path, raw)
    mp.spawn(_worker, args=(world, _free_port(), height, width, raw_path, out_path), nprocs=world, join=True)
    got = np.load(out_path)
    assert np.array_equal(got.view(np.uint64), raw.view(np.uint64))


def test_band_row_map_matches_layout_and_abi():
    """The arithmetic row map the film kernel uses (pt_band_row) against the explicit layout and the C ABI, N = 1..8."""
    import path_tracer_ocaml_amd as P
    from path_tracer_ocaml_amd import distributed as D
    for height in (1, 7, 8, 9, 100, 270, 1080, 2160):
        for world in range(1, 9):
            for band_rows in (1, 8, 32):
                layout = D.band_layout(height, world, band_rows)
                pad = max(len(r) for r in layout)
                idx = D.band_row_index(np.arange(height), world, band_rows, pad)
                assert len(set(idx.tolist())) == height  # injective: no two image rows share a slot
                for r in range(world):
                    params = P.render_params(16, height, 1, 1, band_rows=band_rows, band_first=r, band_step=world)
                    assert P.local_rows(params) == len(layout[r])
                    if world > 1:
                        # local row k of rank r sits at r * pad + k and is image row layout[r][k]
                        assert np.array_equal(idx[layout[r]], r * pad + np.arange(len(layout[r])))
                    if height <= 100:
                        assert [P.global_row(params, k) for k in range(len(layout[r]))] == layout[r].tolist()


def test_band_layout_properties():
    from path_tracer_ocaml_amd import distributed as D
    for h in (1, 32, 33, 1080, 2160):
        for w in (1, 2, 4, 8):
            rows = D.band_layout(h, w)
            allrows = np.concatenate(rows)
            assert sorted(allrows.tolist()) == list(range(h))
            # interleaving balances the load: no rank owns more than one band beyond its share
            assert max(len(r) for r in rows) - min(len(r) for r in rows) <= D.BAND_ROWS
    # BASELINE config 2 on 8 ranks: the busiest rank is within 1 % of the mean
    rows = D.band_layout(1080, 8)
    assert max(len(r) for r in rows) <= 1.01 * 1080 / 8
