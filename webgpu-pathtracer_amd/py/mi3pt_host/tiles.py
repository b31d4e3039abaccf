"""Image tile split across ranks: rows are dealt to ranks in blocks of
`block_rows`, round after round, back and forth (rank r owns block r of even rounds and block nranks - 1 - r of odd ones: a cost
that rises or falls down the image is shared out evenly); every rank keeps a compact [local_rows, width, 4] image.  The
seed of a pixel depends only on its GLOBAL index and the frame (raytrace.wgsl:435-436),
so any split renders identical pixels; one gather of the HDR accumulation buffers plus
this de-interleave reassembles the picture."""
import numpy as np


def local_rows_of(height, rank, nranks, block_rows):
    """Global row indices held by `rank`, in local order -- asked of the LIBRARY (mi3pt_tile_local_rows / mi3pt_tile_global_row, ABI 3: hosts
    must not keep a copy of the deal's formula; round-5 advice: this function used to)."""
    from . import capi
    n = capi.tile_local_rows(height, rank, nranks, block_rows)
    return [capi.tile_global_row(ly, rank, nranks, block_rows) for ly in range(n)]


def deinterleave_rows(parts, height, nranks, block_rows):
    """parts[r]: rank r's (possibly zero-padded) compact image -> the whole image."""
    width = parts[0].shape[1]
    out = np.zeros((height, width, parts[0].shape[2]), parts[0].dtype)
    for r in range(nranks):
        rows = local_rows_of(height, r, nranks, block_rows)
        out[rows] = np.asarray(parts[r])[:len(rows)]
    return out


