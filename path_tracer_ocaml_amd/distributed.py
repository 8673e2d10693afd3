"""Image-band sharding across ranks (one process per GPU) and the single exchange step of the path:
collecting every rank's raw per-pixel sums on rank 0 (RCCL over xGMI: torch.distributed backend "nccl";
"gloo" on CPU for tests).

The reference parallelises over image tiles inside one process (Domainslib pool, integrator.ml:136-151);
across GPUs the unit is a horizontal BAND of `band_rows` (8) image rows, dealt round-robin (band k -> rank
k mod world) so that cheap sky rows and expensive ground rows are spread evenly.  Samples are independent and
the sampler offset depends only on the GLOBAL (x, y, pass) (integrator.ml:98), so the partition does not
change any value.

The step is lean on purpose -- at 1080p over 8 ranks one rank's render share is ~6 ms, so anything rank 0 does
alone afterwards is paid by everybody:

* every buffer is allocated ONCE (:class:`BandGather`); a step allocates nothing;
* each rank renders straight into its send buffer; rank 0 renders straight into slice 0 of the receive buffer;
* the exchange is one group of point-to-point transfers into rank 0 (each peer over its own xGMI link), received
  in place in the layout ``[rank][pad_rows][W][3]``;
* nothing is un-permuted: the film kernel reads that banded layout through an arithmetic row map
  (``ptx_film_resolve_banded_device``, ``pt_band_row`` in csrc/kernels.hip; :func:`band_row_index` is its Python twin).

The same deal and the same film pass are used inside one process by ``ptx_render_multi`` (one host thread per
device, peer-to-peer copies instead of RCCL).
"""
import numpy as np

# 8 rows = the height of the 8x8 pixel tile a wave covers.  Finer bands balance better: at 1080 rows over 8
# ranks, 32-row bands give the busiest rank 160 rows against a mean of 135 (84 % efficiency at best); 8-row
# bands give 136 against 135.
BAND_ROWS = 8


def band_layout(height, world, band_rows=BAND_ROWS):
    """rows[r] = global image rows owned by rank r, in the order ptx_render_raw_device stores them."""
    n_bands = (height + band_rows - 1) // band_rows
    rows = []
    for r in range(world):
        mine = []
        for b in range(r, n_bands, world):
            mine.extend(range(b * band_rows, min((b + 1) * band_rows, height)))
        rows.append(np.asarray(mine, dtype=np.int64))
    return rows


def max_local_rows(height, world, band_rows=BAND_ROWS):
    return max(len(r) for r in band_layout(height, world, band_rows))


def band_row_index(y, world, band_rows, pad_rows):
    """Row of the gathered [world * pad_rows, W, 3] buffer that holds image row y (pt_band_row in csrc/kernels.hip)."""
    y = np.asarray(y)
    if world <= 1:
        return y
    band = y // band_rows
    rank = band % world
    local = (band // world) * band_rows + (y - band * band_rows)
    return rank * pad_rows + local


def ungather(gathered, height, world, band_rows=BAND_ROWS):
    """Reference un-permute of a gathered [world, pad_rows, W, 3] array into image row order (tests and tools only:
    the product's film kernel reads the banded layout in place)."""
    pad = gathered.shape[1]
    flat = gathered.reshape((world * pad,) + tuple(gathered.shape[2:]))
    return flat[band_row_index(np.arange(height), world, band_rows, pad)]


class BandGather:
    """The per-step exchange with every buffer allocated once.

    ``part``      this rank's (pad_rows, W, 3) f64 buffer -- hand ``part.data_ptr()`` to ptx_render_raw_device.
                  On rank 0 it IS slice 0 of ``gathered``.
    ``gathered``  rank 0 only: (world, pad_rows, W, 3), the layout ptx_film_resolve_banded_device reads.
    ``gather()``  one group of sends into rank 0; returns ``gathered`` on rank 0, None elsewhere.
    """

    def __init__(self, height, width, rank, world, device, band_rows=BAND_ROWS, group=None):
        import torch
        self.height, self.width, self.rank, self.world, self.band_rows, self.group = height, width, rank, world, band_rows, group
        self.pad_rows = max_local_rows(height, world, band_rows)
        self.local_rows = len(band_layout(height, world, band_rows)[rank])
        shape = (self.pad_rows, width, 3)
        if rank == 0:
            self.gathered = torch.zeros((world,) + shape, dtype=torch.float64, device=device)
            self.part = self.gathered[0]
        else:
            self.gathered = None
            self.part = torch.zeros(shape, dtype=torch.float64, device=device)
        self._staged = None  # gloo rehearsal with device tensors: host staging buffers, made on first use

    def gather(self):
        if self.world == 1:
            return self.gathered
        import torch
        import torch.distributed as dist
        # gloo moves host memory only: a rehearsal with device tensors stages through the host (never the product
        # path -- the production backend is "nccl" = RCCL, device to device over xGMI)
        staged = dist.get_backend(self.group) == "gloo" and self.part.is_cuda
        if staged and self._staged is None:
            self._staged = torch.empty(((self.world,) if self.rank == 0 else ()) + tuple(self.part.shape), dtype=self.part.dtype)
        if self.rank == 0:
            dst = self._staged if staged else self.gathered
            ops = [dist.P2POp(dist.irecv, dst[r], r, self.group) for r in range(1, self.world)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if staged:
                self.gathered[1:].copy_(dst[1:])
            return self.gathered
        src = self.part
        if staged:
            self._staged.copy_(self.part)
            src = self._staged
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0, self.group)]):
            w.wait()
        return None
